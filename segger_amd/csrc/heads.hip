// Scoring heads: fused cosine-similarity + per-transcript arg-max + assignment
// (prediction) and the triplet margin loss over tx-belongs-bd edges (training).
#include "common.h"
#include "draws.h"

namespace segger {
namespace {

// ---------------------------------------------------------------------------
// Row fragments: a row of C channels is covered by LPC lanes x 8 channels when
// C == 8*LPC (LPC a power of two <= 16), otherwise by one wave striding channels.
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float load1(const T* p);
template <> __device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load1<bf16_t>(const bf16_t* p) { return __uint_as_float((uint32_t)p->v << 16); }
template <> __device__ __forceinline__ float load1<f16_t>(const f16_t* p) {
  return static_cast<float>(__builtin_bit_cast(_Float16, p->v));
}

struct ArgmaxParams {
  const int64_t* indptr; const int32_t* col; const int32_t* eid;
  int64_t n_rows, n_edges;
  const void* zs; int64_t ld_zs;
  const void* zd; int64_t ld_zd;
  int channels;
  float eps; int use_min; float min_sim;
  const int64_t* dst_index;
  float* max_sim; int64_t* max_eid; int64_t* seg_idx; float* sim;
};

template <typename T, int LPC>
__global__ __launch_bounds__(256) void edge_cos_argmax_kernel(ArgmaxParams p) {
  constexpr int RPW = 64 / LPC;                       // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / LPC, gl = lane % LPC;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + grp;
  if (row >= p.n_rows) return;                        // group-uniform; DPP sums stay inside the group
  const T* zs = static_cast<const T*>(p.zs) + row * p.ld_zs + gl * 8;
  const T* zd = static_cast<const T*>(p.zd) + gl * 8;
  float a[8];
  Vec8<T>::load(zs, a);
  float na = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) na = fmaf(a[k], a[k], na);
  na = fmaxf(sqrtf(lane_block_sum<LPC>(na)), p.eps);
  const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
  float best = -INFINITY;
  int64_t best_eid = p.n_edges, best_col = -1;
  for (int64_t e = beg; e < end; ++e) {
    const int64_t j = p.col[e];
    float b[8];
    Vec8<T>::load(zd + j * p.ld_zd, b);
    float dot = 0.f, nb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { dot = fmaf(a[k], b[k], dot); nb = fmaf(b[k], b[k], nb); }
    dot = lane_block_sum<LPC>(dot);
    nb = fmaxf(sqrtf(lane_block_sum<LPC>(nb)), p.eps);
    const float s = dot / (na * nb);
    const int64_t id = p.eid ? (int64_t)p.eid[e] : e;
    if (p.sim && gl == 0) p.sim[id] = s;
    if (s > best) { best = s; best_eid = id; best_col = j; }   // slots of a row are in edge-id order: first max wins
  }
  if (gl == 0) {
    const bool any = best_col >= 0;
    bool valid = any;
    if (p.use_min) valid = valid && (best >= p.min_sim);
    p.max_sim[row] = any ? best : 0.f;
    p.max_eid[row] = best_eid;
    p.seg_idx[row] = valid ? (p.dst_index ? p.dst_index[best_col] : best_col) : -1;
  }
}

// any channel count: one wave per row, lanes stride the channels
template <typename T>
__global__ __launch_bounds__(256) void edge_cos_argmax_generic_kernel(ArgmaxParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= p.n_rows) return;
  const T* zs = static_cast<const T*>(p.zs) + row * p.ld_zs;
  float na = 0.f;
  for (int c = lane; c < p.channels; c += 64) { const float v = load1(zs + c); na = fmaf(v, v, na); }
  na = fmaxf(sqrtf(wave_sum(na)), p.eps);
  const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
  float best = -INFINITY;
  int64_t best_eid = p.n_edges, best_col = -1;
  for (int64_t e = beg; e < end; ++e) {
    const int64_t j = p.col[e];
    const T* zd = static_cast<const T*>(p.zd) + j * p.ld_zd;
    float dot = 0.f, nb = 0.f;
    for (int c = lane; c < p.channels; c += 64) {
      const float x = load1(zs + c), y = load1(zd + c);
      dot = fmaf(x, y, dot); nb = fmaf(y, y, nb);
    }
    dot = wave_sum(dot);
    nb = fmaxf(sqrtf(wave_sum(nb)), p.eps);
    const float s = dot / (na * nb);
    const int64_t id = p.eid ? (int64_t)p.eid[e] : e;
    if (p.sim && lane == 0) p.sim[id] = s;
    if (s > best) { best = s; best_eid = id; best_col = j; }
  }
  if (lane == 0) {
    const bool any = best_col >= 0;
    bool valid = any;
    if (p.use_min) valid = valid && (best >= p.min_sim);
    p.max_sim[row] = any ? best : 0.f;
    p.max_eid[row] = best_eid;
    p.seg_idx[row] = valid ? (p.dst_index ? p.dst_index[best_col] : best_col) : -1;
  }
}

template <typename T>
int launch_argmax(const ArgmaxParams& p, hipStream_t stream) {
  const int C = p.channels;
  auto go = [&](auto lpc_c) {
    constexpr int LPC = decltype(lpc_c)::value;
    const int64_t rows_per_block = 4 * (64 / LPC);
    const int64_t nb = (p.n_rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((edge_cos_argmax_kernel<T, LPC>), dim3((unsigned)nb), dim3(256), 0, stream, p);
  };
  const bool vec_ok = (C % 8 == 0) && aligned16(p.zs) && aligned16(p.zd) &&
                      (p.ld_zs * sizeof(T)) % 16 == 0 && (p.ld_zd * sizeof(T)) % 16 == 0;
  if (vec_ok && C == 8) go(std::integral_constant<int, 1>{});
  else if (vec_ok && C == 16) go(std::integral_constant<int, 2>{});
  else if (vec_ok && C == 32) go(std::integral_constant<int, 4>{});
  else if (vec_ok && C == 64) go(std::integral_constant<int, 8>{});
  else if (vec_ok && C == 128) go(std::integral_constant<int, 16>{});
  else {
    const int64_t nb = (p.n_rows + 3) / 4;
    hipLaunchKernelGGL((edge_cos_argmax_generic_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, p);
  }
  SEGGER_LAUNCH_CHECK("edge_cos_argmax kernel");
  return SEGGER_OK;
}

// ---------------------------------------------------------------------------
// Triplet margin loss
// ---------------------------------------------------------------------------
struct TripletParams {
  const int64_t* src; const int64_t* pos; const int64_t* neg; int64_t n_edges;
  const void* za; int64_t ld_za; const void* zb; int64_t ld_zb;
  int64_t n_a, n_b;      // rows of za / zb: a triplet naming a row outside them is skipped (never dereferenced)
  int channels; float margin, eps;
  float* partial;        // [nblocks]
  float scale;           // bwd: grad_scale / n_edges
  const float* scale_dev; // bwd: optional device multiplier
  void* ga; void* gb;    // bwd: [n_a, C], [n_b, C]; fp32, or the embedding dtype when *_packed
  int ga_packed, gb_packed;
  void* contrib;          // bwd, optional: [n_edges][2][C] fp32 -- (-d/d pos, +d/d neg) per triplet INSTEAD of gb atomics
  const int64_t* pos_indptr; const int32_t* pos_eid;   // bwd, optional: triplets grouped by positive row
  int skip_pos;           // bwd: the positive side of gb is written by triplet_pos_kernel
  void* ga_rows;          // bwd, optional: the anchor's own term of triplet e is STORED into row e of this matrix
};

// gradient accumulation into one row: fp32 atomics, or -- for 16-bit embeddings -- packed 2-channel atomics in the
// embedding's own dtype (global_atomic_pk_add_bf16 / _f16): half the atomics, no fp32 staging buffer, no cast.
// Only for matrices whose rows collect a handful of terms (the transcript side: once as anchor, ~2 as positive /
// negative): a row summing dozens of terms (a boundary embedding) would lose bits, so that side stays fp32.
template <typename T> struct PkAtomic;
template <> struct PkAtomic<float> {
  static __device__ __forceinline__ void add(void*, int64_t, float, float) {}
};
template <> struct PkAtomic<bf16_t> {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void add(void* base, int64_t elem, float a, float b) {
    __builtin_amdgcn_global_atomic_fadd_v2bf16(reinterpret_cast<v2*>(static_cast<bf16_t*>(base) + elem),
                                               __builtin_bit_cast(v2, Vec8<bf16_t>::pack(a, b)));
  }
};
template <> struct PkAtomic<f16_t> {
  typedef _Float16 v2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void add(void* base, int64_t elem, float a, float b) {
    __builtin_amdgcn_global_atomic_fadd_v2f16(reinterpret_cast<v2*>(static_cast<f16_t*>(base) + elem),
                                              __builtin_bit_cast(v2, Vec8<f16_t>::pack(a, b)));
  }
};
template <typename T>
__device__ __forceinline__ void grad_add2(void* base, bool packed, int64_t elem, float a, float b) {
  if (packed) {
    PkAtomic<T>::add(base, elem, a, b);
  } else {
    atomicAdd(static_cast<float*>(base) + elem, a);
    atomicAdd(static_cast<float*>(base) + elem + 1, b);
  }
}

// one wave per edge-slot batch: lanes stride the channels (any C); a wave handles
// kTripletEdgesPerWave edges one after another
constexpr int kTripletEdgesPerBlock = 64;

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void triplet_kernel(TripletParams p) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = p.channels;
  const T* za = static_cast<const T*>(p.za);
  const T* zb = static_cast<const T*>(p.zb);
  // sub-wave groups of 16 lanes, one edge each (C=64 -> 4 channels per lane)
  const int grp = lane >> 4, gl = lane & 15;
  float acc = 0.f;
  const int64_t e_base = (int64_t)blockIdx.x * kTripletEdgesPerBlock;
#pragma unroll 1
  for (int i = wave * 4 + grp; i < kTripletEdgesPerBlock; i += 16) {
    const int64_t e = e_base + i;
    bool ok = e < p.n_edges;                          // group-uniform
    int64_t ia = ok ? p.src[e] : 0, ip = ok ? p.pos[e] : 0, in = ok ? p.neg[e] : 0;
    // ids outside the matrices (a malformed edge_index the deferred validation has not reported yet): the triplet
    // contributes nothing and no row is touched; the mean keeps dividing by n_edges
    if ((uint64_t)ia >= (uint64_t)p.n_a || (uint64_t)ip >= (uint64_t)p.n_b || (uint64_t)in >= (uint64_t)p.n_b) {
      ok = false; ia = ip = in = 0;
    }
    float sp = 0.f, sn = 0.f;
    if (ok)
      for (int c = gl; c < C; c += 16) {
        const float a = load1(za + ia * p.ld_za + c);
        const float dp = a - load1(zb + ip * p.ld_zb + c) + p.eps;
        const float dn = a - load1(zb + in * p.ld_zb + c) + p.eps;
        sp = fmaf(dp, dp, sp); sn = fmaf(dn, dn, sn);
      }
    sp = lane_block_sum<16>(sp);
    sn = lane_block_sum<16>(sn);
    const float dap = sqrtf(sp), dan = sqrtf(sn);
    const float l = dap - dan + p.margin;
    if (!BWD) {
      if (ok && gl == 0) acc += fmaxf(l, 0.f);
    } else if (p.contrib != nullptr && (C & 1) == 0) {
      // z_b side without atomics: every triplet leaves its two rows (zeros when inactive); the caller sums them
      // grouped by boundary (segger_segment_rowsum).  The anchor side keeps its (packed) atomics.
      if (e < p.n_edges) {                             // a skipped (bad-id) triplet still zeroes its two rows
        const bool active = ok && l > 0.f;
        const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
        const float ip_ = active && dap > 0.f ? sc / dap : 0.f;
        const float in_ = active && dan > 0.f ? sc / dan : 0.f;
        float* row = static_cast<float*>(p.contrib) + e * 2 * C;
        for (int c = 2 * gl; c < C; c += 32) {
          const float a0 = load1(za + ia * p.ld_za + c), a1 = load1(za + ia * p.ld_za + c + 1);
          const float dp0 = (a0 - load1(zb + ip * p.ld_zb + c) + p.eps) * ip_;
          const float dp1 = (a1 - load1(zb + ip * p.ld_zb + c + 1) + p.eps) * ip_;
          const float dn0 = (a0 - load1(zb + in * p.ld_zb + c) + p.eps) * in_;
          const float dn1 = (a1 - load1(zb + in * p.ld_zb + c + 1) + p.eps) * in_;
          if (active) grad_add2<T>(p.ga, p.ga_packed, ia * C + c, dp0 - dn0, dp1 - dn1);
          *reinterpret_cast<float2*>(row + c) = float2{-dp0, -dp1};
          *reinterpret_cast<float2*>(row + C + c) = float2{dn0, dn1};
        }
      }
    } else if (p.ga_rows != nullptr) {
      // anchors are the rows themselves (src[e] == e: loss_tx): the anchor term needs no atomic -- row e of ga_rows is
      // WRITTEN (zeros for an inactive / skipped triplet; every row of the matrix is some triplet's), only the
      // positive / negative terms, which land on arbitrary rows, add atomically into ga == gb
      if (e < p.n_edges && e < p.n_a) {
        const bool active = ok && l > 0.f;
        const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
        const float ip_ = active && dap > 0.f ? sc / dap : 0.f;
        const float in_ = active && dan > 0.f ? sc / dan : 0.f;
        for (int c = 2 * gl; c < C; c += 32) {
          const float a0 = load1(za + ia * p.ld_za + c), a1 = load1(za + ia * p.ld_za + c + 1);
          const float dp0 = (a0 - load1(zb + ip * p.ld_zb + c) + p.eps) * ip_;
          const float dp1 = (a1 - load1(zb + ip * p.ld_zb + c + 1) + p.eps) * ip_;
          const float dn0 = (a0 - load1(zb + in * p.ld_zb + c) + p.eps) * in_;
          const float dn1 = (a1 - load1(zb + in * p.ld_zb + c + 1) + p.eps) * in_;
          bool stored = false;
          if constexpr (sizeof(T) == 2) {
            if (p.ga_packed) {
              *reinterpret_cast<uint32_t*>(static_cast<T*>(p.ga_rows) + e * C + c) = Vec8<T>::pack(dp0 - dn0, dp1 - dn1);
              stored = true;
            }
          }
          if (!stored) *reinterpret_cast<float2*>(static_cast<float*>(p.ga_rows) + e * C + c) = float2{dp0 - dn0, dp1 - dn1};
          if (active) {
            if (!p.skip_pos) grad_add2<T>(p.gb, p.gb_packed, ip * C + c, -dp0, -dp1);
            grad_add2<T>(p.gb, p.gb_packed, in * C + c, dn0, dn1);
          }
        }
      }
    } else if (ok && l > 0.f) {
      const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
      const float ip_ = dap > 0.f ? sc / dap : 0.f;
      const float in_ = dan > 0.f ? sc / dan : 0.f;
      if ((C & 1) == 0) {                              // channel pairs (2 gl, 2 gl + 1), + 32 per sweep
        for (int c = 2 * gl; c < C; c += 32) {
          const float a0 = load1(za + ia * p.ld_za + c), a1 = load1(za + ia * p.ld_za + c + 1);
          const float dp0 = (a0 - load1(zb + ip * p.ld_zb + c) + p.eps) * ip_;
          const float dp1 = (a1 - load1(zb + ip * p.ld_zb + c + 1) + p.eps) * ip_;
          const float dn0 = (a0 - load1(zb + in * p.ld_zb + c) + p.eps) * in_;
          const float dn1 = (a1 - load1(zb + in * p.ld_zb + c + 1) + p.eps) * in_;
          grad_add2<T>(p.ga, p.ga_packed, ia * C + c, dp0 - dn0, dp1 - dn1);
          if (!p.skip_pos) grad_add2<T>(p.gb, p.gb_packed, ip * C + c, -dp0, -dp1);
          grad_add2<T>(p.gb, p.gb_packed, in * C + c, dn0, dn1);
        }
      } else {
        float* ga = static_cast<float*>(p.ga);
        float* gb = static_cast<float*>(p.gb);
        for (int c = gl; c < C; c += 16) {
          const float a = load1(za + ia * p.ld_za + c);
          const float dp = (a - load1(zb + ip * p.ld_zb + c) + p.eps) * ip_;
          const float dn = (a - load1(zb + in * p.ld_zb + c) + p.eps) * in_;
          atomicAdd(ga + ia * C + c, dp - dn);
          if (!p.skip_pos) atomicAdd(gb + ip * C + c, -dp);
          atomicAdd(gb + in * C + c, dn);
        }
      }
    }
  }
  if (!BWD) {
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

// Positive side of the boundary gradient without atomics: one wave per positive row j walks the triplets whose
// positive is j (pos_indptr / pos_eid), four at a time (16 lanes each), recomputes their distances and keeps
// -sum_e scale_e * (a_e - z_b[j] + eps) / ||.|| in registers; gb[j, :] is WRITTEN (rows without triplets: zeros).
template <typename T>
__global__ __launch_bounds__(256) void triplet_pos_kernel(TripletParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15;
  const int C = p.channels;
  const int64_t j = (int64_t)blockIdx.x * 4 + wave;
  if (j >= p.n_b) return;                               // wave-uniform
  const T* za = static_cast<const T*>(p.za);
  const T* zb = static_cast<const T*>(p.zb);
  float* gb = static_cast<float*>(p.gb);
  const int64_t beg = p.pos_indptr[j], end = p.pos_indptr[j + 1];
  const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
  // channels c = gl + 16*k: up to 8 per lane in registers (C <= 128), more in further sweeps
  for (int c0 = 0; c0 < C; c0 += 128) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int64_t s0 = beg; s0 < end; s0 += 4) {
      const int64_t s = s0 + grp;
      bool ok = s < end;
      const int64_t e = ok ? (int64_t)p.pos_eid[s] : 0;
      int64_t ia = ok ? p.src[e] : 0, in = ok ? p.neg[e] : 0;
      if ((uint64_t)ia >= (uint64_t)p.n_a || (uint64_t)in >= (uint64_t)p.n_b) { ok = false; ia = in = 0; }
      float sp = 0.f, sn = 0.f;
      for (int c = gl; c < C; c += 16) {
        const float a = load1(za + ia * p.ld_za + c);
        const float dp = a - load1(zb + j * p.ld_zb + c) + p.eps;
        const float dn = a - load1(zb + in * p.ld_zb + c) + p.eps;
        sp = fmaf(dp, dp, sp); sn = fmaf(dn, dn, sn);
      }
      sp = lane_block_sum<16>(sp);
      sn = lane_block_sum<16>(sn);
      const float dap = sqrtf(sp), dan = sqrtf(sn);
      const bool active = ok && (dap - dan + p.margin > 0.f) && dap > 0.f;
      const float w = active ? sc / dap : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c0 + gl + 16 * k;
        if (c < C) acc[k] -= (load1(za + ia * p.ld_za + c) - load1(zb + j * p.ld_zb + c) + p.eps) * w;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = acc[k];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int c = c0 + gl + 16 * k;
      if (grp == 0 && c < C) gb[j * C + c] = v;
    }
  }
}

// The whole backward of the tx-belongs-bd triplets in ONE walk over the groups (triplets grouped by positive row j, as
// triplet_pos_kernel), for triplets whose anchors are all different (a transcript lies in at most one boundary):
//   * anchor row: STORED (one 4-byte pair or 8 bytes per lane; rows that are no anchor keep the caller's zeros) instead
//     of 32 packed atomics per triplet;
//   * positive row j: accumulated in registers over the group, added to grad_b ONCE per boundary;
//   * negative row: fp32 atomics (uniformly sampled: uncontended).
// Every row of z_a / z_b is read once per triplet (the two-kernel route reads them twice).  Lane gl of a 16-lane group
// owns the channel pairs 2 gl + 32 k, k < C / 32 (C % 32 == 0, C <= 128): 4-byte loads for 16-bit embeddings.
template <typename T> __device__ __forceinline__ void load2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void load2<float>(const float* p, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(p);
  a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void load2<bf16_t>(const bf16_t* p, float& a, float& b) {
  Vec8<bf16_t>::unpack2(*reinterpret_cast<const uint32_t*>(p), a, b);
}
template <> __device__ __forceinline__ void load2<f16_t>(const f16_t* p, float& a, float& b) {
  Vec8<f16_t>::unpack2(*reinterpret_cast<const uint32_t*>(p), a, b);
}

template <typename T, int KP>
__global__ __launch_bounds__(256) void triplet_grouped_kernel(TripletParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15;
  constexpr int C = 32 * KP;
  // one workgroup per boundary: its 16 lane groups take 16 triplets per iteration (a boundary has ~40: the walk is a
  // chain of dependent loads -- triplet id -> row ids -> rows -- and one wave per boundary left 10 such rounds in series)
  const int64_t j = blockIdx.x;
  const T* za = static_cast<const T*>(p.za);
  const T* zb = static_cast<const T*>(p.zb);
  float* gb = static_cast<float*>(p.gb);
  const int64_t beg = p.pos_indptr[j] + 4 * wave, end = p.pos_indptr[j + 1];
  if (beg >= end) return;                               // wave-uniform
  const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
  float pj[KP][2], acc[KP][2];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    load2(zb + j * p.ld_zb + 2 * gl + 32 * k, pj[k][0], pj[k][1]);
    acc[k][0] = 0.f; acc[k][1] = 0.f;
  }
  for (int64_t s0 = beg; s0 < end; s0 += 16) {
    const int64_t s = s0 + grp;
    bool ok = s < end;
    const int64_t e = ok ? (int64_t)p.pos_eid[s] : 0;
    int64_t ia = ok ? p.src[e] : 0, in = ok ? p.neg[e] : 0;
    // a triplet whose own positive is not this group's row (padding: pos = -1) or with ids outside the matrices is skipped
    if ((ok && p.pos[e] != j) || (uint64_t)ia >= (uint64_t)p.n_a || (uint64_t)in >= (uint64_t)p.n_b) { ok = false; ia = in = 0; }
    float dp[KP][2], dn[KP][2];
    float sp = 0.f, sn = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      float a0, a1, n0, n1;
      load2(za + ia * p.ld_za + 2 * gl + 32 * k, a0, a1);
      load2(zb + in * p.ld_zb + 2 * gl + 32 * k, n0, n1);
      dp[k][0] = a0 - pj[k][0] + p.eps; dp[k][1] = a1 - pj[k][1] + p.eps;
      dn[k][0] = a0 - n0 + p.eps;       dn[k][1] = a1 - n1 + p.eps;
      sp = fmaf(dp[k][0], dp[k][0], sp); sp = fmaf(dp[k][1], dp[k][1], sp);
      sn = fmaf(dn[k][0], dn[k][0], sn); sn = fmaf(dn[k][1], dn[k][1], sn);
    }
    sp = lane_block_sum<16>(sp);
    sn = lane_block_sum<16>(sn);
    const float dap = sqrtf(sp), dan = sqrtf(sn);
    if (ok && dap - dan + p.margin > 0.f) {              // group-uniform
      const float ip_ = dap > 0.f ? sc / dap : 0.f;
      const float in_ = dan > 0.f ? sc / dan : 0.f;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const float p0 = dp[k][0] * ip_, p1 = dp[k][1] * ip_, n0 = dn[k][0] * in_, n1 = dn[k][1] * in_;
        acc[k][0] -= p0; acc[k][1] -= p1;
        const int64_t ea = ia * C + 2 * gl + 32 * k;
        bool stored = false;
        if constexpr (sizeof(T) == 2) {
          if (p.ga_packed) {
            *reinterpret_cast<uint32_t*>(static_cast<T*>(p.ga) + ea) = Vec8<T>::pack(p0 - n0, p1 - n1);
            stored = true;
          }
        }
        if (!stored) *reinterpret_cast<float2*>(static_cast<float*>(p.ga) + ea) = float2{p0 - n0, p1 - n1};
        atomicAdd(gb + in * C + 2 * gl + 32 * k, n0);
        atomicAdd(gb + in * C + 2 * gl + 32 * k + 1, n1);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v = acc[k][q];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (grp == 0) atomicAdd(gb + j * C + 2 * gl + 32 * k + q, v);     // (negatives of other groups land here too)
    }
  }
}

// ---- BCE segmentation head (lightning_model.py:190-207): BCEWithLogits over the dot-product logits of the positive
// edges (label 1) and the sampled negative edges (label 0), mean over both:
//     loss = 1/(2n) sum_e [ softplus(-<a_e, p_e>) + softplus(<a_e, n_e>) ]
//     d/d lp = (sigmoid(lp) - 1) / (2n),  d/d ln = sigmoid(ln) / (2n)
// Same gather / scatter structure as the triplet kernels: 16 lanes per edge, channel pairs 2 gl + 32 k.
__device__ __forceinline__ float softplus_f(float x) {          // log(1 + e^x), stable
  return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bce_edge_kernel(TripletParams p) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = p.channels;                              // even (checked by the host)
  const T* za = static_cast<const T*>(p.za);
  const T* zb = static_cast<const T*>(p.zb);
  const int grp = lane >> 4, gl = lane & 15;
  float acc = 0.f;
  const int64_t e_base = (int64_t)blockIdx.x * kTripletEdgesPerBlock;
#pragma unroll 1
  for (int i = wave * 4 + grp; i < kTripletEdgesPerBlock; i += 16) {
    const int64_t e = e_base + i;
    bool ok = e < p.n_edges;                          // group-uniform
    int64_t ia = ok ? p.src[e] : 0, ip = ok ? p.pos[e] : 0, in = ok ? p.neg[e] : 0;
    if ((uint64_t)ia >= (uint64_t)p.n_a || (uint64_t)ip >= (uint64_t)p.n_b || (uint64_t)in >= (uint64_t)p.n_b) {
      ok = false; ia = ip = in = 0;
    }
    float lp = 0.f, ln = 0.f;
    for (int c = 2 * gl; c < C; c += 32) {
      float a0, a1, p0, p1, n0, n1;
      load2(za + ia * p.ld_za + c, a0, a1);
      load2(zb + ip * p.ld_zb + c, p0, p1);
      load2(zb + in * p.ld_zb + c, n0, n1);
      lp = fmaf(a0, p0, fmaf(a1, p1, lp));
      ln = fmaf(a0, n0, fmaf(a1, n1, ln));
    }
    lp = lane_block_sum<16>(lp);
    ln = lane_block_sum<16>(ln);
    if (!BWD) {
      if (ok && gl == 0) acc += softplus_f(-lp) + softplus_f(ln);
    } else if (ok) {
      const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
      const float dlp = (sigmoid_f(lp) - 1.0f) * sc, dln = sigmoid_f(ln) * sc;
      for (int c = 2 * gl; c < C; c += 32) {
        float a0, a1, p0, p1, n0, n1;
        load2(za + ia * p.ld_za + c, a0, a1);
        load2(zb + ip * p.ld_zb + c, p0, p1);
        load2(zb + in * p.ld_zb + c, n0, n1);
        grad_add2<T>(p.ga, p.ga_packed, ia * C + c, dlp * p0 + dln * n0, dlp * p1 + dln * n1);
        grad_add2<T>(p.gb, p.gb_packed, ip * C + c, dlp * a0, dlp * a1);
        grad_add2<T>(p.gb, p.gb_packed, in * C + c, dln * a0, dln * a1);
      }
    }
  }
  if (!BWD) {
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

// backward with unique anchors, one workgroup per positive row (as triplet_grouped_kernel): anchor rows stored, the
// positive row summed in registers and added once, negatives by fp32 atomics
template <typename T, int KP>
__global__ __launch_bounds__(256) void bce_grouped_kernel(TripletParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15;
  constexpr int C = 32 * KP;
  const int64_t j = blockIdx.x;
  const T* za = static_cast<const T*>(p.za);
  const T* zb = static_cast<const T*>(p.zb);
  float* gb = static_cast<float*>(p.gb);
  const int64_t beg = p.pos_indptr[j] + 4 * wave, end = p.pos_indptr[j + 1];
  if (beg >= end) return;                               // wave-uniform
  const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
  float pj[KP][2], acc[KP][2];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    load2(zb + j * p.ld_zb + 2 * gl + 32 * k, pj[k][0], pj[k][1]);
    acc[k][0] = 0.f; acc[k][1] = 0.f;
  }
  for (int64_t s0 = beg; s0 < end; s0 += 16) {
    const int64_t s = s0 + grp;
    bool ok = s < end;
    const int64_t e = ok ? (int64_t)p.pos_eid[s] : 0;
    int64_t ia = ok ? p.src[e] : 0, in = ok ? p.neg[e] : 0;
    if ((ok && p.pos[e] != j) || (uint64_t)ia >= (uint64_t)p.n_a || (uint64_t)in >= (uint64_t)p.n_b) { ok = false; ia = in = 0; }
    float av[KP][2], nv[KP][2];
    float lp = 0.f, ln = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      load2(za + ia * p.ld_za + 2 * gl + 32 * k, av[k][0], av[k][1]);
      load2(zb + in * p.ld_zb + 2 * gl + 32 * k, nv[k][0], nv[k][1]);
      lp = fmaf(av[k][0], pj[k][0], fmaf(av[k][1], pj[k][1], lp));
      ln = fmaf(av[k][0], nv[k][0], fmaf(av[k][1], nv[k][1], ln));
    }
    lp = lane_block_sum<16>(lp);
    ln = lane_block_sum<16>(ln);
    if (ok) {                                            // group-uniform
      const float dlp = (sigmoid_f(lp) - 1.0f) * sc, dln = sigmoid_f(ln) * sc;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        acc[k][0] = fmaf(dlp, av[k][0], acc[k][0]); acc[k][1] = fmaf(dlp, av[k][1], acc[k][1]);
        const float g0 = dlp * pj[k][0] + dln * nv[k][0], g1 = dlp * pj[k][1] + dln * nv[k][1];
        const int64_t ea = ia * C + 2 * gl + 32 * k;
        bool stored = false;
        if constexpr (sizeof(T) == 2) {
          if (p.ga_packed) {
            *reinterpret_cast<uint32_t*>(static_cast<T*>(p.ga) + ea) = Vec8<T>::pack(g0, g1);
            stored = true;
          }
        }
        if (!stored) *reinterpret_cast<float2*>(static_cast<float*>(p.ga) + ea) = float2{g0, g1};
        atomicAdd(gb + in * C + 2 * gl + 32 * k, dln * av[k][0]);
        atomicAdd(gb + in * C + 2 * gl + 32 * k + 1, dln * av[k][1]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v = acc[k][q];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (grp == 0) atomicAdd(gb + j * C + 2 * gl + 32 * k + q, v);
    }
  }
}

// C == 64 fast path: a lane owns 4 consecutive channels (one 8-byte load per row for 16-bit embeddings, 16 bytes
// for fp32), 16 lanes per triplet, 4 triplets per wave-iteration; the backward issues two packed (or four fp32)
// atomics per row and lane.
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&f)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&f)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&f)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float (&f)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  Vec8<f16_t>::unpack(v.x, f[0], f[1]);
  Vec8<f16_t>::unpack(v.y, f[2], f[3]);
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void triplet_c64_kernel(TripletParams p) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int C = 64;
  const T* za = static_cast<const T*>(p.za);
  const T* zb = static_cast<const T*>(p.zb);
  const int grp = lane >> 4, c0 = (lane & 15) * 4;
  float acc = 0.f;
  const int64_t e_base = (int64_t)blockIdx.x * kTripletEdgesPerBlock;
#pragma unroll 1
  for (int i = wave * 4 + grp; i < kTripletEdgesPerBlock; i += 16) {
    const int64_t e = e_base + i;
    bool ok = e < p.n_edges;                          // group-uniform
    int64_t ia = ok ? p.src[e] : 0, ip = ok ? p.pos[e] : 0, in = ok ? p.neg[e] : 0;
    // ids outside the matrices (a malformed edge_index the deferred validation has not reported yet): the triplet
    // contributes nothing and no row is touched; the mean keeps dividing by n_edges
    if ((uint64_t)ia >= (uint64_t)p.n_a || (uint64_t)ip >= (uint64_t)p.n_b || (uint64_t)in >= (uint64_t)p.n_b) {
      ok = false; ia = ip = in = 0;
    }
    float a[4], pp[4], nn[4], dp[4], dn[4];
    load4(za + ia * p.ld_za + c0, a);
    load4(zb + ip * p.ld_zb + c0, pp);
    load4(zb + in * p.ld_zb + c0, nn);
    float sp = 0.f, sn = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dp[k] = a[k] - pp[k] + p.eps; dn[k] = a[k] - nn[k] + p.eps;
      sp = fmaf(dp[k], dp[k], sp); sn = fmaf(dn[k], dn[k], sn);
    }
    sp = lane_block_sum<16>(sp);
    sn = lane_block_sum<16>(sn);
    const float dap = sqrtf(sp), dan = sqrtf(sn);
    const float l = dap - dan + p.margin;
    if (!BWD) {
      if (ok && (lane & 15) == 0) acc += fmaxf(l, 0.f);
    } else if (ok && l > 0.f) {
      const float sc = p.scale_dev ? p.scale * p.scale_dev[0] : p.scale;
      const float ip_ = dap > 0.f ? sc / dap : 0.f;
      const float in_ = dan > 0.f ? sc / dan : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { dp[k] *= ip_; dn[k] *= in_; }
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        grad_add2<T>(p.ga, p.ga_packed, ia * C + c0 + k, dp[k] - dn[k], dp[k + 1] - dn[k + 1]);
        grad_add2<T>(p.gb, p.gb_packed, ip * C + c0 + k, -dp[k], -dp[k + 1]);
        grad_add2<T>(p.gb, p.gb_packed, in * C + c0 + k, dn[k], dn[k + 1]);
      }
    }
  }
  if (!BWD) {
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

__global__ __launch_bounds__(256) void triplet_finish_kernel(const float* __restrict__ partial, int64_t n, float inv_n,
                                                            float* __restrict__ loss) {
  __shared__ float wsum[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_n;
}

__global__ __launch_bounds__(256) void triplet_sample_kernel(SampleParams p) { triplet_sample_body(p, blockIdx.x); }

// ---- MetricLoss on sampled triplets: 16 lanes per node, 4 nodes per wave-iteration ----------------------------------
struct MetricParams {
  const void* z; int64_t ld; int64_t n; int channels;
  const int64_t* pos; const int64_t* neg; const float* dpos; const float* dneg; const float* w; float eps;
  float* partial; const float* scale_dev; float* gz;
};

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void metric_kernel(MetricParams p) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15;
  const int C = p.channels;
  const T* z = static_cast<const T*>(p.z);
  float acc = 0.f;
  const int64_t base = (int64_t)blockIdx.x * kTripletEdgesPerBlock;
#pragma unroll 1
  for (int k = wave * 4 + grp; k < kTripletEdgesPerBlock; k += 16) {
    const int64_t i = base + k;
    bool ok = i < p.n;
    int64_t ip = ok ? p.pos[i] : 0, in = ok ? p.neg[i] : 0;
    const float w = ok ? p.w[i] : 0.f;
    if ((uint64_t)ip >= (uint64_t)p.n || (uint64_t)in >= (uint64_t)p.n || w == 0.f) { ok = false; ip = in = 0; }
    const int64_t ii = ok ? i : 0;
    float sxx = 0.f, spp = 0.f, snn = 0.f, sxp = 0.f, sxn = 0.f;
    for (int c = gl; c < C; c += 16) {
      const float x = load1(z + ii * p.ld + c), yp = load1(z + ip * p.ld + c), yn = load1(z + in * p.ld + c);
      sxx = fmaf(x, x, sxx); spp = fmaf(yp, yp, spp); snn = fmaf(yn, yn, snn);
      sxp = fmaf(x, yp, sxp); sxn = fmaf(x, yn, sxn);
    }
    sxx = lane_block_sum<16>(sxx); spp = lane_block_sum<16>(spp); snn = lane_block_sum<16>(snn);
    sxp = lane_block_sum<16>(sxp); sxn = lane_block_sum<16>(sxn);
    const float nx = sqrtf(sxx), np_ = sqrtf(spp), nn_ = sqrtf(snn);
    const float cx = fmaxf(nx, p.eps), cp = fmaxf(np_, p.eps), cn = fmaxf(nn_, p.eps);
    const float cos_p = sxp / (cx * cp), cos_n = sxn / (cx * cn);
    const float rp = cos_p - (1.f - (ok ? p.dpos[i] : 0.f)), rn = cos_n - (1.f - (ok ? p.dneg[i] : 0.f));
    if (!BWD) {
      if (ok && gl == 0) acc += w * (rp * rp + rn * rn);
    } else if (ok) {
      const float sc = p.scale_dev ? p.scale_dev[0] : 1.f;
      const float gp = 2.f * w * rp * sc, gn = 2.f * w * rn * sc;         // d loss / d cos
      // d cos(x, y) / d x = y / (cx cy) - [|x| > eps] cos x / |x|^2     (a clamped norm is a constant)
      const float ax = (nx > p.eps ? (gp * cos_p + gn * cos_n) / sxx : 0.f);
      const float bp = (np_ > p.eps ? gp * cos_p / spp : 0.f), bn = (nn_ > p.eps ? gn * cos_n / snn : 0.f);
      const float ip_ = gp / (cx * cp), in_ = gn / (cx * cn);
      for (int c = gl; c < C; c += 16) {
        const float x = load1(z + i * p.ld + c), yp = load1(z + ip * p.ld + c), yn = load1(z + in * p.ld + c);
        atomicAdd(p.gz + i * C + c, ip_ * yp + in_ * yn - ax * x);
        atomicAdd(p.gz + ip * C + c, ip_ * x - bp * yp);
        atomicAdd(p.gz + in * C + c, in_ * x - bn * yn);
      }
    }
  }
  if (!BWD) {
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

int64_t triplet_blocks(int64_t n_edges) { return (n_edges + kTripletEdgesPerBlock - 1) / kTripletEdgesPerBlock; }

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_edge_cos_argmax(const segger_edge_argmax_args* a, segger_stream_t stream) {
  SEGGER_REQUIRE(a != nullptr, "segger_edge_cos_argmax: args is NULL");
  const segger_csr& g = a->by_src;
  SEGGER_REQUIRE(g.n_rows >= 0 && g.n_edges >= 0 && g.n_cols >= 0, "segger_edge_cos_argmax: negative size");
  SEGGER_REQUIRE(g.n_rows < 0x7fffffffLL && g.n_edges < 0x7fffffffLL, "segger_edge_cos_argmax: batch too large");
  SEGGER_REQUIRE(a->channels > 0, "segger_edge_cos_argmax: channels must be positive");
  if (g.n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(g.indptr && (g.n_edges == 0 || g.col), "segger_edge_cos_argmax: NULL graph array");
  SEGGER_REQUIRE(a->z_src && (g.n_edges == 0 || a->z_dst), "segger_edge_cos_argmax: NULL embedding");
  SEGGER_REQUIRE(a->ld_zs >= a->channels && a->ld_zd >= a->channels, "segger_edge_cos_argmax: ld < channels");
  SEGGER_REQUIRE(a->max_sim && a->max_eid && a->seg_idx, "segger_edge_cos_argmax: NULL output");
  ArgmaxParams p{g.indptr, g.col, g.eid, g.n_rows, g.n_edges, a->z_src, a->ld_zs, a->z_dst, a->ld_zd, a->channels,
                 a->eps, a->use_min_similarity, a->min_similarity, a->dst_index, a->max_sim, a->max_eid, a->seg_idx, a->sim};
  switch (a->dtype) {
    case SEGGER_F32:  return launch_argmax<float>(p, (hipStream_t)stream);
    case SEGGER_BF16: return launch_argmax<bf16_t>(p, (hipStream_t)stream);
    case SEGGER_F16:  return launch_argmax<f16_t>(p, (hipStream_t)stream);
    default: set_error("segger_edge_cos_argmax: unknown dtype %d", a->dtype); return SEGGER_EINVAL;
  }
}

extern "C" int64_t segger_triplet_partial_count(int64_t n_edges) { return n_edges > 0 ? triplet_blocks(n_edges) : 0; }

extern "C" size_t segger_triplet_workspace_bytes(int64_t n_edges) {
  return (size_t)(n_edges > 0 ? triplet_blocks(n_edges) : 1) * sizeof(float) + 16;
}

static int triplet_common(const segger_triplet_args* a, bool bwd, hipStream_t stream) {
  SEGGER_REQUIRE(a != nullptr, "segger_triplet: args is NULL");
  SEGGER_REQUIRE(a->n_edges >= 0 && a->channels > 0, "segger_triplet: bad sizes");
  SEGGER_REQUIRE(a->n_edges < 0x7fffffffLL * kTripletEdgesPerBlock, "segger_triplet: too many edges");
  if (a->n_edges == 0) {
    // torch: mean over zero elements is NaN; segger never calls the loss with no edges
    // (lightning_model.py:173 guards num_bd <= 1 only).  We return 0.
    if (!bwd && a->loss) SEGGER_HIP(hipMemsetAsync(a->loss, 0, sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(a->src && a->pos && a->neg && a->z_a && a->z_b, "segger_triplet: NULL input");
  SEGGER_REQUIRE(a->ld_za >= a->channels && a->ld_zb >= a->channels, "segger_triplet: ld < channels");
  SEGGER_REQUIRE(a->n_a > 0 && a->n_b > 0, "segger_triplet: n_a / n_b (rows of z_a / z_b) must be positive");
  const int64_t nb = triplet_blocks(a->n_edges);
  TripletParams p{a->src, a->pos, a->neg, a->n_edges, a->z_a, a->ld_za, a->z_b, a->ld_zb, a->n_a, a->n_b, a->channels, a->margin, a->eps,
                  static_cast<float*>(a->workspace), 0.f, nullptr, a->grad_a, a->grad_b,
                  a->grad_a_packed, a->grad_b_packed, bwd ? a->contrib : nullptr,
                  bwd ? a->pos_indptr : nullptr, bwd ? a->pos_eid : nullptr, 0, bwd ? a->grad_a_rows : nullptr};
  if (!bwd) {
    const size_t need = segger_triplet_workspace_bytes(a->n_edges);
    if (!a->workspace || a->workspace_bytes < need) {
      set_error("segger_triplet_fwd: workspace %zu < %zu bytes", a->workspace_bytes, need);
      return SEGGER_EWORKSPACE;
    }
  } else {
    SEGGER_REQUIRE(a->grad_a && (a->grad_b || a->contrib), "segger_triplet_bwd: NULL gradient buffer");
    SEGGER_REQUIRE(!a->contrib || (a->channels % 2 == 0 && a->grad_a != a->grad_b),
                   "segger_triplet_bwd: contrib needs an even channel count and a separate z_b");
    SEGGER_REQUIRE(!(a->grad_a_packed || a->grad_b_packed) || (a->dtype != SEGGER_F32 && a->channels % 2 == 0),
                   "segger_triplet_bwd: packed gradient buffers need a 16-bit dtype and an even channel count");
    SEGGER_REQUIRE(!(a->grad_a == a->grad_b) || a->grad_a_packed == a->grad_b_packed,
                   "segger_triplet_bwd: one shared gradient buffer cannot be both packed and fp32");
    SEGGER_REQUIRE(!a->grad_a_rows || (a->channels % 2 == 0 && !a->contrib && a->loss_kind == SEGGER_LOSS_TRIPLET &&
                                       a->n_edges == a->n_a && ((uintptr_t)a->grad_a_rows % 8) == 0),
                   "segger_triplet_bwd: grad_a_rows needs an even channel count, one triplet per row of z_a, no contrib");
    p.scale = a->grad_scale / (float)a->n_edges;
    p.scale_dev = a->grad_scale_dev;
    if (p.pos_indptr) {
      SEGGER_REQUIRE(p.pos_eid && a->grad_b && a->grad_b != a->grad_a && !a->grad_b_packed && !a->contrib,
                     "segger_triplet_bwd: pos_indptr needs pos_eid and a separate fp32 grad_b (no contrib)");
      p.skip_pos = 1;
    }
  }
  dim3 grid((unsigned)nb), block(256);
  if (a->loss_kind == SEGGER_LOSS_BCE) {
    // BCE head: mean over the 2 n logits; 4-byte pair loads
    const size_t es1 = a->dtype == SEGGER_F32 ? 4 : 2;
    SEGGER_REQUIRE(a->channels % 2 == 0 && (a->ld_za * es1) % 4 == 0 && (a->ld_zb * es1) % 4 == 0 &&
                       ((uintptr_t)a->z_a % 8) == 0 && ((uintptr_t)a->z_b % 8) == 0,
                   "segger_triplet (BCE): even channel count and 8-byte aligned rows");
    SEGGER_REQUIRE(!a->contrib, "segger_triplet (BCE): contrib is a triplet-loss option");
    p.scale *= 0.5f;
    const bool grouped = bwd && a->anchor_unique && p.pos_indptr && a->channels % 32 == 0 && a->channels <= 128 &&
                         ((uintptr_t)a->grad_a % 8) == 0;
    p.skip_pos = 0;                                     // (the two-kernel positive route is a triplet-loss option)
#define BCE(T)                                                                                                    \
    do {                                                                                                          \
      if (grouped) {                                                                                              \
        const dim3 gg((unsigned)a->n_b);                                                                          \
        switch (a->channels / 32) {                                                                               \
          case 1: hipLaunchKernelGGL((bce_grouped_kernel<T, 1>), gg, block, 0, stream, p); break;                 \
          case 2: hipLaunchKernelGGL((bce_grouped_kernel<T, 2>), gg, block, 0, stream, p); break;                 \
          case 3: hipLaunchKernelGGL((bce_grouped_kernel<T, 3>), gg, block, 0, stream, p); break;                 \
          default: hipLaunchKernelGGL((bce_grouped_kernel<T, 4>), gg, block, 0, stream, p); break;                \
        }                                                                                                         \
      } else if (bwd) hipLaunchKernelGGL((bce_edge_kernel<T, true>), grid, block, 0, stream, p);                  \
      else hipLaunchKernelGGL((bce_edge_kernel<T, false>), grid, block, 0, stream, p);                            \
    } while (0)
    SEGGER_REQUIRE(a->n_b < 0x7fffffffLL, "segger_triplet (BCE): too many rows in z_b");
    switch (a->dtype) {
      case SEGGER_F32:  BCE(float); break;
      case SEGGER_BF16: BCE(bf16_t); break;
      case SEGGER_F16:  BCE(f16_t); break;
      default: set_error("segger_triplet: unknown dtype %d", a->dtype); return SEGGER_EINVAL;
    }
#undef BCE
    SEGGER_LAUNCH_CHECK("bce kernels");
    if (!bwd && a->loss) {
      hipLaunchKernelGGL(triplet_finish_kernel, dim3(1), dim3(256), 0, stream, p.partial, nb, 0.5f / (float)a->n_edges, a->loss);
      SEGGER_LAUNCH_CHECK("triplet_finish_kernel");
    }
    return SEGGER_OK;
  }
  if (bwd && a->anchor_unique && p.pos_indptr) {
    // one walk over the groups does everything (see triplet_grouped_kernel); grad_b is accumulated into
    const size_t es0 = a->dtype == SEGGER_F32 ? 4 : 2;
    SEGGER_REQUIRE(a->channels % 32 == 0 && a->channels <= 128 && (a->ld_za * es0) % 4 == 0 && (a->ld_zb * es0) % 4 == 0 &&
                       ((uintptr_t)a->z_a % 8) == 0 && ((uintptr_t)a->z_b % 8) == 0 && ((uintptr_t)a->grad_a % 8) == 0,
                   "segger_triplet_bwd: anchor_unique needs C in {32, 64, 96, 128} and 8-byte aligned rows");
    SEGGER_REQUIRE(a->n_b < 0x7fffffffLL, "segger_triplet_bwd: too many rows in z_b");
    const dim3 ggrid((unsigned)a->n_b);
#define GROUPED(T)                                                                                         \
    switch (a->channels / 32) {                                                                            \
      case 1: hipLaunchKernelGGL((triplet_grouped_kernel<T, 1>), ggrid, block, 0, stream, p); break;       \
      case 2: hipLaunchKernelGGL((triplet_grouped_kernel<T, 2>), ggrid, block, 0, stream, p); break;       \
      case 3: hipLaunchKernelGGL((triplet_grouped_kernel<T, 3>), ggrid, block, 0, stream, p); break;       \
      default: hipLaunchKernelGGL((triplet_grouped_kernel<T, 4>), ggrid, block, 0, stream, p); break;      \
    }
    switch (a->dtype) {
      case SEGGER_F32:  GROUPED(float); break;
      case SEGGER_BF16: GROUPED(bf16_t); break;
      case SEGGER_F16:  GROUPED(f16_t); break;
      default: set_error("segger_triplet: unknown dtype %d", a->dtype); return SEGGER_EINVAL;
    }
#undef GROUPED
    SEGGER_LAUNCH_CHECK("triplet_grouped_kernel");
    return SEGGER_OK;
  }
  // C == 64 with rows that admit 4-channel vector loads -> the vectorised kernel
  const size_t es = a->dtype == SEGGER_F32 ? 4 : 2;
  const bool c64 = a->channels == 64 && (a->ld_za * es) % (4 * es) == 0 && (a->ld_zb * es) % (4 * es) == 0 &&
                   ((uintptr_t)a->z_a % (4 * es)) == 0 && ((uintptr_t)a->z_b % (4 * es)) == 0;
#define LAUNCH(T)                                                                                \
  do {                                                                                           \
    if (bwd && p.skip_pos)   /* first: it WRITES grad_b; the per-triplet kernel then adds the negatives */            \
      hipLaunchKernelGGL((triplet_pos_kernel<T>), dim3((unsigned)((a->n_b + 3) / 4)), block, 0, stream, p);   \
    if (c64 && !bwd) {   /* backward: the lane-strided pair layout of triplet_kernel keeps each atomic   */ \
      /* instruction on consecutive dwords; the 4-channel layout measured 60 % slower (0.71 vs 0.44 ms) */   \
      hipLaunchKernelGGL((triplet_c64_kernel<T, false>), grid, block, 0, stream, p);             \
    } else {                                                                                     \
      if (bwd) hipLaunchKernelGGL((triplet_kernel<T, true>), grid, block, 0, stream, p);         \
      else     hipLaunchKernelGGL((triplet_kernel<T, false>), grid, block, 0, stream, p);        \
    }                                                                                            \
  } while (0)
  switch (a->dtype) {
    case SEGGER_F32:  LAUNCH(float); break;
    case SEGGER_BF16: LAUNCH(bf16_t); break;
    case SEGGER_F16:  LAUNCH(f16_t); break;
    default: set_error("segger_triplet: unknown dtype %d", a->dtype); return SEGGER_EINVAL;
  }
#undef LAUNCH
  SEGGER_LAUNCH_CHECK("triplet_kernel");
  if (!bwd && a->loss) {      // loss == NULL: the per-block partial sums stay in the workspace (segger_loss_combine_partials_fwd)
    hipLaunchKernelGGL(triplet_finish_kernel, dim3(1), dim3(256), 0, stream, p.partial, nb, 1.0f / (float)a->n_edges, a->loss);
    SEGGER_LAUNCH_CHECK("triplet_finish_kernel");
  }
  return SEGGER_OK;
}

// ---- the weighted sum of the step's losses and its gradient, as one tiny launch each way --------------------------------
namespace segger {
namespace {
__global__ void loss_combine_kernel(const float* __restrict__ raw, const float* __restrict__ a, const float* __restrict__ b,
                                    int n, float* __restrict__ out, const float* __restrict__ gout, float* __restrict__ graw) {
  if (threadIdx.x != 0) return;
  if (gout == nullptr) {
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
      const float t = raw[i] * a[i];
      out[i] = t;
      total = fmaf(t, b[i], total);
    }
    out[n] = total;
  } else {
    for (int i = 0; i < n; ++i) graw[i] = (gout[n] * b[i] + gout[i]) * a[i];
  }
}
}  // namespace
}  // namespace segger

namespace segger {
namespace {
struct CombineParts { const float* partial[16]; int64_t n_partial[16]; float scale[16]; };
// one workgroup: term i = scale[i] * (sum of its partial sums, in a fixed order), then the combination of loss_combine_kernel
__global__ __launch_bounds__(256) void loss_combine_parts_kernel(CombineParts c, const float* __restrict__ a,
                                                                const float* __restrict__ b, int n, float* __restrict__ out) {
  __shared__ float wsum[4];
  __shared__ float raw[16];
  for (int i = 0; i < n; ++i) {
    float s = 0.f;
    for (int64_t k = threadIdx.x; k < c.n_partial[i]; k += 256) s += c.partial[i][k];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) raw[i] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * c.scale[i];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
      const float t = raw[i] * a[i];
      out[i] = t;
      total = fmaf(t, b[i], total);
    }
    out[n] = total;
  }
}
}  // namespace
}  // namespace segger

extern "C" int segger_loss_combine_partials_fwd(const float* const* partial, const int64_t* n_partial, const float* scale,
                                                const float* a, const float* b, int32_t n, float* out,
                                                segger_stream_t stream) {
  SEGGER_REQUIRE(n > 0 && n <= 16 && partial && n_partial && scale && a && b && out,
                 "segger_loss_combine_partials_fwd: 1..16 terms, no NULL pointer");
  CombineParts c{};
  for (int i = 0; i < n; ++i) {
    SEGGER_REQUIRE(n_partial[i] >= 0 && (n_partial[i] == 0 || partial[i]), "segger_loss_combine_partials_fwd: bad term %d", i);
    c.partial[i] = partial[i]; c.n_partial[i] = n_partial[i]; c.scale[i] = scale[i];
  }
  hipLaunchKernelGGL(loss_combine_parts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, c, a, b, n, out);
  SEGGER_LAUNCH_CHECK("loss_combine_parts_kernel");
  return SEGGER_OK;
}

extern "C" int segger_loss_combine_fwd(const float* raw, const float* a, const float* b, int32_t n, float* out,
                                       segger_stream_t stream) {
  SEGGER_REQUIRE(n > 0 && n <= 16 && raw && a && b && out, "segger_loss_combine_fwd: 1..16 terms, no NULL pointer");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, raw, a, b, n, out, nullptr, nullptr);
  SEGGER_LAUNCH_CHECK("loss_combine_kernel");
  return SEGGER_OK;
}
extern "C" int segger_loss_combine_bwd(const float* grad_out, const float* a, const float* b, int32_t n, float* grad_raw,
                                       segger_stream_t stream) {
  SEGGER_REQUIRE(n > 0 && n <= 16 && grad_out && a && b && grad_raw, "segger_loss_combine_bwd: 1..16 terms, no NULL pointer");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nullptr, a, b, n, nullptr, grad_out,
                     grad_raw);
  SEGGER_LAUNCH_CHECK("loss_combine_kernel");
  return SEGGER_OK;
}

extern "C" int segger_triplet_fwd(const segger_triplet_args* a, segger_stream_t stream) {
  return triplet_common(a, false, (hipStream_t)stream);
}
extern "C" int segger_triplet_bwd(const segger_triplet_args* a, segger_stream_t stream) {
  return triplet_common(a, true, (hipStream_t)stream);
}

extern "C" int segger_triplet_sample(const int64_t* lab, int64_t n, int32_t n_clusters, const float* cdf_pos,
                                     const float* cdf_neg, const int64_t* counts, const int64_t* offsets,
                                     const int64_t* members, const float* uniforms, uint64_t seed, const uint64_t* seed_dev,
                                     const float* dists, int64_t* pos, int64_t* neg, float* d_pos, float* d_neg,
                                     segger_stream_t stream) {
  SEGGER_REQUIRE(n >= 0 && n_clusters > 0, "segger_triplet_sample: bad sizes");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(lab && cdf_pos && cdf_neg && counts && offsets && members && pos && neg,
                 "segger_triplet_sample: NULL pointer");
  SEGGER_REQUIRE(!dists == !d_pos && !d_pos == !d_neg, "segger_triplet_sample: dists, d_pos and d_neg go together");
  const uint64_t mixed = splitmix64(seed);
  SampleParams p{lab, n, n_clusters, cdf_pos, cdf_neg, counts, offsets, members, uniforms,
                 (uint32_t)mixed, (uint32_t)(mixed >> 32), seed, seed_dev, dists, pos, neg, d_pos, d_neg};
  hipLaunchKernelGGL(triplet_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  SEGGER_LAUNCH_CHECK("triplet_sample_kernel");
  return SEGGER_OK;
}

namespace segger {
__global__ __launch_bounds__(256) void sample_negatives_kernel(NegParams p) { sample_negatives_body(p, blockIdx.x); }
}  // namespace segger

extern "C" int segger_sample_negatives(const int64_t* pos, int64_t n, int64_t n_b, const int64_t* n_b_dev, uint64_t seed,
                                       const uint64_t* seed_dev, int64_t* neg, segger_stream_t stream) {
  SEGGER_REQUIRE(n >= 0 && n_b >= 0, "segger_sample_negatives: negative size");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(pos && neg, "segger_sample_negatives: NULL pointer");
  hipLaunchKernelGGL(sample_negatives_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     NegParams{pos, n, n_b, n_b_dev, seed, seed_dev, neg});
  SEGGER_LAUNCH_CHECK("sample_negatives_kernel");
  return SEGGER_OK;
}

static int metric_common(const void* z, int64_t ld_z, int64_t n, int32_t channels, int32_t dtype, const int64_t* pos,
                         const int64_t* neg, const float* d_pos, const float* d_neg, const float* w, float eps,
                         float* loss, void* workspace, size_t workspace_bytes, const float* scale_dev, float* grad_z,
                         bool bwd, hipStream_t stream) {
  SEGGER_REQUIRE(n >= 0 && channels > 0 && ld_z >= channels, "segger_metric: bad sizes");
  if (n == 0) {
    if (!bwd && loss) SEGGER_HIP(hipMemsetAsync(loss, 0, sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(z && pos && neg && d_pos && d_neg && w, "segger_metric: NULL input");
  const int64_t nb = triplet_blocks(n);
  MetricParams p{z, ld_z, n, channels, pos, neg, d_pos, d_neg, w, eps, static_cast<float*>(workspace), scale_dev, grad_z};
  if (!bwd) {
    const size_t need = segger_triplet_workspace_bytes(n);
    if (!workspace || workspace_bytes < need) {
      set_error("segger_metric_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
      return SEGGER_EWORKSPACE;
    }
  } else {
    SEGGER_REQUIRE(grad_z != nullptr, "segger_metric_bwd: grad_z is NULL");
  }
  dim3 grid((unsigned)nb), block(256);
#define LAUNCH_M(T)                                                                                \
  do {                                                                                             \
    if (bwd) hipLaunchKernelGGL((metric_kernel<T, true>), grid, block, 0, stream, p);              \
    else     hipLaunchKernelGGL((metric_kernel<T, false>), grid, block, 0, stream, p);             \
  } while (0)
  switch (dtype) {
    case SEGGER_F32:  LAUNCH_M(float); break;
    case SEGGER_BF16: LAUNCH_M(bf16_t); break;
    case SEGGER_F16:  LAUNCH_M(f16_t); break;
    default: set_error("segger_metric: unknown dtype %d", dtype); return SEGGER_EINVAL;
  }
#undef LAUNCH_M
  SEGGER_LAUNCH_CHECK("metric_kernel");
  if (!bwd && loss) {         // loss == NULL: partial sums only, as segger_triplet_fwd
    hipLaunchKernelGGL(triplet_finish_kernel, dim3(1), dim3(256), 0, stream, p.partial, nb, 1.0f, loss);
    SEGGER_LAUNCH_CHECK("triplet_finish_kernel");
  }
  return SEGGER_OK;
}

extern "C" int segger_metric_fwd(const void* z, int64_t ld_z, int64_t n, int32_t channels, int32_t dtype, const int64_t* pos,
                                 const int64_t* neg, const float* d_pos, const float* d_neg, const float* w, float eps,
                                 float* loss, void* workspace, size_t workspace_bytes, segger_stream_t stream) {
  return metric_common(z, ld_z, n, channels, dtype, pos, neg, d_pos, d_neg, w, eps, loss, workspace, workspace_bytes,
                       nullptr, nullptr, false, (hipStream_t)stream);
}
extern "C" int segger_metric_bwd(const void* z, int64_t ld_z, int64_t n, int32_t channels, int32_t dtype, const int64_t* pos,
                                 const int64_t* neg, const float* d_pos, const float* d_neg, const float* w, float eps,
                                 const float* grad_scale_dev, float* grad_z, segger_stream_t stream) {
  return metric_common(z, ld_z, n, channels, dtype, pos, neg, d_pos, d_neg, w, eps, nullptr, nullptr, 0, grad_scale_dev,
                       grad_z, true, (hipStream_t)stream);
}
