// segger_stage: many small "copy a prefix, fill the rest by a formula" jobs in ONE launch.
//
// A captured training step (hipGraph) reads static, padded buffers; every batch has to be written into them first.
// Done with framework copies / fills that is ~95 launches of 3-5 us per 1 M-edge batch -- more than the GATv2 forward
// of the batch.  Here the host describes all of them as segments and one grid (blockIdx.y = segment) executes them.
#include "common.h"

namespace segger {

constexpr int kStageMaxSegs = 56;        // 56 x 64 B by value in the kernel arguments (< 4 KiB)

struct StageBatch {
  segger_stage_seg s[kStageMaxSegs];
};

template <typename T>
__device__ __forceinline__ int64_t load_int(const void* p, int64_t i) { return (int64_t)static_cast<const T*>(p)[i]; }

__device__ __forceinline__ int64_t load_elem(const void* p, int64_t i, int bytes) {
  switch (bytes) {
    case 1:  return load_int<uint8_t>(p, i);
    case 2:  return load_int<uint16_t>(p, i);
    case 4:  return load_int<int32_t>(p, i);
    default: return load_int<int64_t>(p, i);
  }
}

__device__ __forceinline__ void store_elem(void* p, int64_t i, int bytes, int64_t v) {
  switch (bytes) {
    case 1:  static_cast<uint8_t*>(p)[i] = (uint8_t)v; break;
    case 2:  static_cast<uint16_t*>(p)[i] = (uint16_t)v; break;
    case 4:  static_cast<int32_t*>(p)[i] = (int32_t)v; break;
    default: static_cast<int64_t*>(p)[i] = v; break;
  }
}

__global__ __launch_bounds__(256) void stage_kernel(StageBatch b) {
  const segger_stage_seg& g = b.s[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g.n_total; i += stride) {
    int64_t v;
    if (i < g.n_copy) {
      v = load_elem(g.src, i, g.src_bytes) + g.copy_add;
    } else {
      const int64_t k = i - g.n_copy;
      switch (g.fill) {
        case SEGGER_FILL_TILE: v = load_elem(g.src, k % g.a, g.src_bytes) + g.copy_add; break;
        case SEGGER_FILL_DIV:  v = g.a + k / g.b; break;
        case SEGGER_FILL_MOD:  v = g.a + k % g.b; break;
        case SEGGER_FILL_RAMP: { const int64_t t = (k + 1) * g.b; v = g.a + (t < g.c ? t : g.c); break; }
        default:               v = g.a; break;
      }
    }
    store_elem(g.dst, i, g.dst_bytes, v);
  }
}

}  // namespace segger

using namespace segger;

extern "C" int segger_stage(const segger_stage_seg* segs, int32_t n_segs, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_segs >= 0, "segger_stage: negative segment count");
  SEGGER_REQUIRE(n_segs == 0 || segs != nullptr, "segger_stage: segs is NULL");
  for (int32_t s0 = 0; s0 < n_segs; s0 += kStageMaxSegs) {
    StageBatch b;
    const int n = (n_segs - s0 < kStageMaxSegs) ? (n_segs - s0) : kStageMaxSegs;
    int64_t longest = 0;
    for (int i = 0; i < n; ++i) {
      const segger_stage_seg& g = segs[s0 + i];
      auto size_ok = [](int32_t v) { return v == 1 || v == 2 || v == 4 || v == 8; };
      SEGGER_REQUIRE(g.n_copy >= 0 && g.n_total >= g.n_copy, "segger_stage: segment %d: need 0 <= n_copy <= n_total", s0 + i);
      SEGGER_REQUIRE(size_ok(g.dst_bytes) && size_ok(g.src_bytes), "segger_stage: segment %d: element size must be 1, 2, 4 or 8", s0 + i);
      SEGGER_REQUIRE(g.n_total == 0 || g.dst != nullptr, "segger_stage: segment %d: dst is NULL", s0 + i);
      SEGGER_REQUIRE(g.n_copy == 0 || g.src != nullptr, "segger_stage: segment %d: src is NULL", s0 + i);
      SEGGER_REQUIRE(g.fill >= SEGGER_FILL_CONST && g.fill <= SEGGER_FILL_RAMP, "segger_stage: segment %d: unknown fill", s0 + i);
      if (g.n_total > g.n_copy) {
        if (g.fill == SEGGER_FILL_TILE)
          SEGGER_REQUIRE(g.src != nullptr && g.a >= 1, "segger_stage: segment %d: TILE needs src and a period >= 1", s0 + i);
        if (g.fill == SEGGER_FILL_DIV || g.fill == SEGGER_FILL_MOD)
          SEGGER_REQUIRE(g.b >= 1, "segger_stage: segment %d: DIV / MOD need b >= 1", s0 + i);
      }
      b.s[i] = g;
      if (g.n_total > longest) longest = g.n_total;
    }
    if (longest == 0) continue;
    int64_t gx = (longest + 1023) / 1024;                 // ~4 elements per thread on the longest segment
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(stage_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, stream, b);
    SEGGER_LAUNCH_CHECK("stage_kernel");
  }
  return SEGGER_OK;
}

// ---- many small 2-D transposes of 16-bit matrices in one launch -------------------------------------------------------
// The data gradient dX = dY W runs on the projection kernel with W^T as its weight: every optimizer step the compute-
// dtype copies of ~12 weight matrices are re-transposed.  One launch (blockIdx.y = matrix) instead of one per matrix.
namespace segger {
constexpr int kTrMaxSegs = 32;
struct TrBatch { segger_transpose_seg s[kTrMaxSegs]; };

__global__ __launch_bounds__(256) void transpose_many_kernel(TrBatch b) {
  __shared__ uint16_t tile[32][33];
  const segger_transpose_seg& g = b.s[blockIdx.y];
  const uint16_t* __restrict__ src = static_cast<const uint16_t*>(g.src);
  uint16_t* __restrict__ dst = static_cast<uint16_t*>(g.dst);
  const int tr = (g.rows + 31) / 32, tc = (g.cols + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8 threads
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      tile[ty + 8 * i][tx] = (r < g.rows && c < g.cols) ? src[(int64_t)r * g.cols + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;                      // dst is [cols, rows]
      if (c < g.cols && r < g.rows) dst[(int64_t)c * g.rows + r] = tile[tx][ty + 8 * i];
    }
    __syncthreads();
  }
}
}  // namespace segger

extern "C" int segger_transpose_many(const segger_transpose_seg* segs, int32_t n_segs, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_segs >= 0, "segger_transpose_many: negative segment count");
  SEGGER_REQUIRE(n_segs == 0 || segs != nullptr, "segger_transpose_many: segs is NULL");
  for (int32_t s0 = 0; s0 < n_segs; s0 += kTrMaxSegs) {
    TrBatch b;
    const int n = (n_segs - s0 < kTrMaxSegs) ? (n_segs - s0) : kTrMaxSegs;
    int most = 0;
    for (int i = 0; i < n; ++i) {
      const segger_transpose_seg& g = segs[s0 + i];
      SEGGER_REQUIRE(g.rows >= 0 && g.cols >= 0 && (g.rows == 0 || g.cols == 0 || (g.src && g.dst)),
                     "segger_transpose_many: segment %d: bad shape or NULL pointer", s0 + i);
      b.s[i] = g;
      const int tiles = ((g.rows + 31) / 32) * ((g.cols + 31) / 32);
      if (tiles > most) most = tiles;
    }
    if (most == 0) continue;
    if (most > 64) most = 64;
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)most, (unsigned)n), dim3(256), 0, stream, b);
    SEGGER_LAUNCH_CHECK("transpose_many_kernel");
  }
  return SEGGER_OK;
}

// ---- compute-dtype copies of the fp32 master weights, all of them in one launch (segger_pack_refresh) --------------------
// After an optimizer step every projection's fp32 weight is cast into its window of a row-stacked 16-bit buffer, copied
// transposed into the buffer the data gradients stream, and its bias copied into the stacked fp32 bias: torch did this
// as two multi-tensor copies + segger_transpose_many; here blockIdx.y = segment, 32 x 32 tiles through LDS.
namespace segger {
constexpr int kPackMaxSegs = 64;
struct PackBatch { segger_pack_seg s[kPackMaxSegs]; int dtype; };

template <typename T>
__device__ __forceinline__ uint16_t cast16(float v);
template <> __device__ __forceinline__ uint16_t cast16<bf16_t>(float v) { return (uint16_t)(Vec8<bf16_t>::pack(v, 0.f) & 0xffffu); }
template <> __device__ __forceinline__ uint16_t cast16<f16_t>(float v) { return (uint16_t)(Vec8<f16_t>::pack(v, 0.f) & 0xffffu); }

template <typename T>
__global__ __launch_bounds__(256) void pack_refresh_kernel(PackBatch b) {
  __shared__ uint16_t tile[32][33];
  const segger_pack_seg& g = b.s[blockIdx.y];
  const float* __restrict__ src = g.src;
  if (g.dst_f32) {                                      // bias: fp32 -> fp32, rows elements
    float* __restrict__ d = static_cast<float*>(g.dst);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < g.rows; i += gridDim.x * 256) d[i] = src[i];
    return;
  }
  uint16_t* __restrict__ dst = static_cast<uint16_t*>(g.dst);
  uint16_t* __restrict__ dst_t = static_cast<uint16_t*>(g.dst_t);
  const int tr = (g.rows + 31) / 32, tc = (g.cols + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8 threads
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      uint16_t v = 0;
      if (r < g.rows && c < g.cols) {
        v = cast16<T>(src[(int64_t)r * g.cols + c]);
        dst[(int64_t)r * g.cols + c] = v;
      }
      tile[ty + 8 * i][tx] = v;
    }
    if (dst_t) {                                         // (uniform over the workgroup)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;                    // dst_t is [cols, ld_t], this matrix at its column window
        if (c < g.cols && r < g.rows) dst_t[(int64_t)c * g.ld_t + r] = tile[tx][ty + 8 * i];
      }
      __syncthreads();
    }
  }
}
}  // namespace segger

extern "C" int segger_pack_refresh(const segger_pack_seg* segs, int32_t n_segs, int32_t dtype, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_segs >= 0, "segger_pack_refresh: negative segment count");
  SEGGER_REQUIRE(n_segs == 0 || segs != nullptr, "segger_pack_refresh: segs is NULL");
  SEGGER_REQUIRE(dtype == SEGGER_BF16 || dtype == SEGGER_F16, "segger_pack_refresh: bf16 / f16 destinations");
  for (int32_t s0 = 0; s0 < n_segs; s0 += kPackMaxSegs) {
    PackBatch b;
    b.dtype = dtype;
    const int n = (n_segs - s0 < kPackMaxSegs) ? (n_segs - s0) : kPackMaxSegs;
    int most = 0;
    for (int i = 0; i < n; ++i) {
      const segger_pack_seg& g = segs[s0 + i];
      SEGGER_REQUIRE(g.rows >= 0 && g.cols >= 0 && (g.rows == 0 || g.cols == 0 || (g.src && g.dst)),
                     "segger_pack_refresh: segment %d: bad shape or NULL pointer", s0 + i);
      SEGGER_REQUIRE(!g.dst_t || g.ld_t >= g.rows, "segger_pack_refresh: segment %d: ld_t < rows", s0 + i);
      b.s[i] = g;
      const int tiles = g.dst_f32 ? (g.rows + 255) / 256 : ((g.rows + 31) / 32) * ((g.cols + 31) / 32);
      if (tiles > most) most = tiles;
    }
    if (most == 0) continue;
    if (most > 64) most = 64;
    if (dtype == SEGGER_BF16)
      hipLaunchKernelGGL(pack_refresh_kernel<bf16_t>, dim3((unsigned)most, (unsigned)n), dim3(256), 0, stream, b);
    else
      hipLaunchKernelGGL(pack_refresh_kernel<f16_t>, dim3((unsigned)most, (unsigned)n), dim3(256), 0, stream, b);
    SEGGER_LAUNCH_CHECK("pack_refresh_kernel");
  }
  return SEGGER_OK;
}
