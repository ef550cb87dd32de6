// Positional embedder, stage 1: per-graph min / max of node positions.
// Replaces the Python loop over graphs (one boolean mask + two reductions + a
// host sync per graph) of reference src/segger/models/ist_encoder.py:66-73.
#include "common.h"

namespace segger {
namespace {

// order-preserving float atomics on plain global words initialised to +inf / -inf
// (-0.0f compares >= 0 but its bit pattern is INT_MIN: canonicalise it to +0 before choosing the integer view)
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  v += 0.f;
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else          atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  v += 0.f;
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else          atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void minmax_init_kernel(float* mins, float* maxs, int64_t n2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) { mins[i] = INFINITY; maxs[i] = -INFINITY; }
}

// nodes per thread: 4 for tile batches (more workgroups, shorter per-thread chains), 16 from 256 k nodes on (fewer
// same-address atomics)
constexpr int kNodesSmall = 4, kNodesLarge = 16;
constexpr int64_t kNodesLargeFrom = 262144;

struct MM { float lx, ly, hx, hy; };

__device__ __forceinline__ void flush(float* mins, float* maxs, int64_t g, const MM& m) {
  atomic_min_f32(mins + 2 * g, m.lx); atomic_min_f32(mins + 2 * g + 1, m.ly);
  atomic_max_f32(maxs + 2 * g, m.hx); atomic_max_f32(maxs + 2 * g + 1, m.hy);
}

// Graph ids of a PyG Batch are sorted, so a block's contiguous chunk of nodes
// nearly always belongs to one graph: lanes accumulate privately, a wave whose
// lanes all hold the same graph reduces with shuffles and issues 4 atomics.
template <int kNodesPerThread>
__global__ __launch_bounds__(256) void segment_minmax_kernel(const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                                            int64_t n, int64_t n_graphs, float* mins, float* maxs) {
  const int64_t base = (int64_t)blockIdx.x * (256 * kNodesPerThread);
  int64_t cur = -1;
  MM m{INFINITY, INFINITY, -INFINITY, -INFINITY};
  // all of a thread's loads first (clamped indices, straight-line): taken one node at a time behind the branches
  // below they were 16 dependent round trips -- 25 us for 55 k nodes and for 1 M alike
  int64_t gs[kNodesPerThread];
  float2 ps[kNodesPerThread];
#pragma unroll
  for (int i = 0; i < kNodesPerThread; ++i) {
    int64_t v = base + (int64_t)i * 256 + threadIdx.x;
    const bool in = v < n;
    if (!in) v = n - 1;                            // (n > 0: checked on the host)
    gs[i] = in ? (batch ? batch[v] : 0) : -1;
    ps[i] = *reinterpret_cast<const float2*>(pos + 2 * v);
  }
#pragma unroll
  for (int i = 0; i < kNodesPerThread; ++i) {
    const int64_t g = gs[i];
    if (g < 0 || g >= n_graphs) continue;          // past the end / ignored, like an unmatched mask in the reference loop
    const float2 p = ps[i];
    if (g != cur) {
      if (cur >= 0) flush(mins, maxs, cur, m);
      cur = g;
      m = MM{p.x, p.y, p.x, p.y};
    } else {
      m.lx = fminf(m.lx, p.x); m.ly = fminf(m.ly, p.y);
      m.hx = fmaxf(m.hx, p.x); m.hy = fmaxf(m.hy, p.y);
    }
  }
  const int cur32 = (int)cur;                      // n_graphs < 2^31 (checked on the host)
  const int first = __shfl(cur32, 0, 64);
  const bool uniform = __all(cur32 == first);
  // per wave: one (graph, box) when its lanes agree; then per block: one flush when its four waves agree too (every
  // atomic of a tile batch lands on the same few addresses -- same-address atomics serialise)
  __shared__ int wg[4];
  __shared__ MM wm[4];
  const int wave = threadIdx.x >> 6;
  if (uniform && first >= 0) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m.lx = fminf(m.lx, __shfl_xor(m.lx, off, 64)); m.ly = fminf(m.ly, __shfl_xor(m.ly, off, 64));
      m.hx = fmaxf(m.hx, __shfl_xor(m.hx, off, 64)); m.hy = fmaxf(m.hy, __shfl_xor(m.hy, off, 64));
    }
  } else if (!uniform && cur >= 0) {
    flush(mins, maxs, cur, m);
  }
  if ((threadIdx.x & 63) == 0) { wg[wave] = uniform ? first : -1; wm[wave] = m; }   // -1: nothing left to flush for this wave
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool same = wg[0] >= 0 && wg[1] == wg[0] && wg[2] == wg[0] && wg[3] == wg[0];
    if (same) {
      MM a = wm[0];
      for (int k = 1; k < 4; ++k) {
        a.lx = fminf(a.lx, wm[k].lx); a.ly = fminf(a.ly, wm[k].ly); a.hx = fmaxf(a.hx, wm[k].hx); a.hy = fmaxf(a.hy, wm[k].hy);
      }
      flush(mins, maxs, wg[0], a);
    } else {
      for (int k = 0; k < 4; ++k)
        if (wg[k] >= 0) flush(mins, maxs, wg[k], wm[k]);
    }
  }
}

// graphs without nodes keep (0, 0) as in the reference (ist_encoder.py:67-68)
__global__ __launch_bounds__(256) void minmax_fixup_kernel(float* mins, float* maxs, int64_t n2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2 && mins[i] == INFINITY && maxs[i] == -INFINITY) { mins[i] = 0.f; maxs[i] = 0.f; }
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_segment_minmax_ex(const float* pos, const int64_t* batch, int64_t n, int64_t n_graphs,
                                        float* mins, float* maxs, int32_t flags, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && n_graphs >= 0, "segger_segment_minmax: negative size");
  SEGGER_REQUIRE(n_graphs < 0x7fffffffLL, "segger_segment_minmax: too many graphs");
  if (n_graphs == 0) return SEGGER_OK;
  SEGGER_REQUIRE(mins && maxs, "segger_segment_minmax: NULL output");
  SEGGER_REQUIRE(n == 0 || pos, "segger_segment_minmax: pos is NULL");
  SEGGER_REQUIRE((reinterpret_cast<uintptr_t>(pos) & 7u) == 0, "segger_segment_minmax: pos must be 8-byte aligned");
  const int64_t n2 = 2 * n_graphs;
  const unsigned gb = (unsigned)((n2 + 255) / 256);
  if (!(flags & SEGGER_MINMAX_INITIALISED))
    hipLaunchKernelGGL(minmax_init_kernel, dim3(gb), dim3(256), 0, stream, mins, maxs, n2);
  if (n > 0) {
    const int64_t per = 256 * (n >= kNodesLargeFrom ? kNodesLarge : kNodesSmall);
    const int64_t nb = (n + per - 1) / per;
    SEGGER_REQUIRE(nb < 0x7fffffffLL, "segger_segment_minmax: too many nodes");
    if (n >= kNodesLargeFrom)
      hipLaunchKernelGGL(segment_minmax_kernel<kNodesLarge>, dim3((unsigned)nb), dim3(256), 0, stream, pos, batch, n, n_graphs, mins, maxs);
    else
      hipLaunchKernelGGL(segment_minmax_kernel<kNodesSmall>, dim3((unsigned)nb), dim3(256), 0, stream, pos, batch, n, n_graphs, mins, maxs);
  }
  if (!(flags & SEGGER_MINMAX_KEEP_EMPTY))
    hipLaunchKernelGGL(minmax_fixup_kernel, dim3(gb), dim3(256), 0, stream, mins, maxs, n2);
  SEGGER_LAUNCH_CHECK("segment_minmax kernels");
  return SEGGER_OK;
}

extern "C" int segger_segment_minmax(const float* pos, const int64_t* batch, int64_t n, int64_t n_graphs,
                                     float* mins, float* maxs, segger_stream_t stream) {
  return segger_segment_minmax_ex(pos, batch, n, n_graphs, mins, maxs, 0, stream);
}
