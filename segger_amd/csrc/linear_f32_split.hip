// fp32-storage projections on the bf16 matrix pipe: every fp32 operand as the sum of THREE bf16 numbers (8 + 8 + 8
// mantissa bits = fp32's 24), the product as the six partial products that matter,
//     x w  =  xh wh + (xh wm + xm wh) + (xh wl + xl wh + xm wm)  +  O(2^-24 |x w|),
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Per 32 x 32 output tile and 16 k that
// is 6 instructions of 8 passes on the 2.5 PFLOP/s bf16 pipe where the exact-fp32 form v_mfma_f32_32x32x2_f32
// (csrc/linear_f32.hip, 157 TFLOP/s) needs 8 instructions of 16 passes: 0.375 of the matrix time, and the product (MFMA-bound
// as exact fp32: 1.17 ms for 1M x 128 -> 384) comes within reach of its 2 GB of fp32 traffic.  Measured (tools/
// bench_f32_split.py): 0.77 ms, error relative to sum |x||w| 3.2e-7 against 3.5e-7 for the exact kernel.
// NOT bit-identical to fp32 arithmetic (the dropped terms are below 2^-24 relative to |x||w|; the exact kernels stay the
// parity mode): opt-in, ops.F32_SPLIT.
//
//   segger_linear_fwd_f32_split:  Y[n, M] = X[n, K] W^T + b,  W given as three bf16 planes [3][M][K] (hi, mid, lo)
//     K = 128 (forward of the 128 -> M projections), or K = 384 with M = 128 (their data gradient dX = dY W: W^T planes)
#include <cstdlib>
#include "common.h"

namespace segger {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float pk2 __attribute__((ext_vector_type(2)));          // a register pair for v_pk_mul / v_pk_add / v_pk_fma_f32

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A workgroup barrier that orders LDS traffic only: this wave's LDS operations have completed (lgkmcnt(0)), then s_barrier.
// __syncthreads() is a release fence as well -- s_waitcnt vmcnt(0): every global load in flight (the prefetch these kernels
// live on) and every output store would have to drain at each barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");                         // (compiler: no memory access moves across)
  __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct SplitParams {
  const float* x; int64_t ldx;
  const bf16_t* w3;          // [3][m_out][K]
  const float* bias;
  float* y; int64_t ldy;
  int64_t n_rows;
  int m_out;
  const float* rowbias; const int32_t* rowidx; int64_t ld_rb;     // optional: y[row, :] += rowbias[rowidx[row], :] (fp32 table)
  const float* gate; int64_t ld_gate; int gate_kind;              // optional: y *= act'(gate) (1 = GELU, 2 = SiLU)
};

// 8 consecutive floats -> their hi / mid / lo bf16 parts (round to nearest each time: the remainders are exact in fp32)
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
#ifdef EXP_FS_NOSPLIT        // bounding build: no VALU split (planes = raw words)
  hi = __builtin_bit_cast(u32x4, a); mid = __builtin_bit_cast(u32x4, b); lo = hi ^ mid;
  return;
#endif
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = Vec8<bf16_t>::pack(v[2 * i], v[2 * i + 1]);
    float h0, h1;
    Vec8<bf16_t>::unpack2(h[i], h0, h1);
    const float r0 = v[2 * i] - h0, r1 = v[2 * i + 1] - h1;
    m[i] = Vec8<bf16_t>::pack(r0, r1);
    float m0, m1;
    Vec8<bf16_t>::unpack2(m[i], m0, m1);
    l[i] = Vec8<bf16_t>::pack(r0 - m0, r1 - m1);
  }
  hi = u32x4{h[0], h[1], h[2], h[3]}; mid = u32x4{m[0], m[1], m[2], m[3]}; lo = u32x4{l[0], l[1], l[2], l[3]};
}

constexpr int kKS = 128;                       // k extent held as fragments at a time
constexpr int kCH = 64;                        // output columns per staged W chunk
constexpr int kWS = kKS * 2 + 16;              // LDS row stride of a staged plane (bytes): conflict-free ds_read_b128

// A workgroup (4 waves) owns 128 rows; a wave keeps the split fragments of its 32 rows x 128 k in registers (96 VGPRs).
//   K == 128: chunks of 64 output columns stream through LDS (three planes: 52 KB), one accumulator pair per chunk;
//   K == 384 (M == 128): k slices outermost, the whole [32, 128] output of a wave stays in accumulators.
#ifndef SEGGER_FS_K384_WAVES
#define SEGGER_FS_K384_WAVES 1            // K = 384: one wave per SIMD (512 registers): next slice's rows AND next W chunk in flight
#endif
template <int K>
__global__ __launch_bounds__(256, K == 128 ? 2 : SEGGER_FS_K384_WAVES) void linear_f32_split_kernel(SplitParams p) {
  constexpr int NS = K / kKS;                  // k slices
  constexpr int NACC = NS == 1 ? 2 : 4;        // 32-column accumulator tiles alive
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * kCH * kWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
  int64_t row = row0 + r;
  const bool row_ok = row < p.n_rows;
  if (!row_ok) row = p.n_rows - 1;             // clamp: loaded, never stored
  const int M = p.m_out;

  u32x4 xh[8], xm[8], xl[8];
  auto load_x = [&](int slice) {
    const float* xr = p.x + row * p.ldx + slice * kKS + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * s), b = *reinterpret_cast<const f32x4*>(xr + 16 * s + 4);
      split8(a, b, xh[s], xm[s], xl[s]);
    }
  };
  // planes of W rows [c0, c0 + 64), k in [slice * 128, +128) -> LDS (3 x 64 rows x 16 pieces of 16 bytes)
  auto stage_w = [&](int c0, int slice) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int piece = tid + 256 * i;                       // 0 .. 3071
      const int plane = piece >> 10, wrow = (piece >> 4) & 63, wcol = piece & 15;
      const u32x4 v = *reinterpret_cast<const u32x4*>(p.w3 + ((int64_t)plane * M + c0 + wrow) * K + slice * kKS + wcol * 8);
      *reinterpret_cast<u32x4*>(lds + (plane * kCH + wrow) * kWS + wcol * 16) = v;
    }
  };
  // the same in two halves: global -> registers (in flight under the previous chunk's MFMAs AND output stores: vmcnt is
  // in order, so a chunk staged only after the stores were issued waits for them to drain -- 6 store drains per block),
  // registers -> LDS behind the barrier
  u32x4 wreg[12];
  auto w_fetch = [&](int c0, int slice) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int piece = tid + 256 * i;
      const int plane = piece >> 10, wrow = (piece >> 4) & 63, wcol = piece & 15;
      wreg[i] = *reinterpret_cast<const u32x4*>(p.w3 + ((int64_t)plane * M + c0 + wrow) * K + slice * kKS + wcol * 8);
    }
  };
  auto w_commit = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int piece = tid + 256 * i;
      const int plane = piece >> 10, wrow = (piece >> 4) & 63, wcol = piece & 15;
      *reinterpret_cast<u32x4*>(lds + (plane * kCH + wrow) * kWS + wcol * 16) = wreg[i];
    }
  };
  // both 32-column tiles of the staged chunk, their (dependent) MFMA chains interleaved
  auto chunk_mma = [&](f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int off = r * kWS + (16 * s + 8 * h) * 2;
      const u32x4 wh0 = *reinterpret_cast<const u32x4*>(lds + off), wh1 = *reinterpret_cast<const u32x4*>(lds + off + 32 * kWS);
      const u32x4 wm0 = *reinterpret_cast<const u32x4*>(lds + kCH * kWS + off);
      const u32x4 wm1 = *reinterpret_cast<const u32x4*>(lds + kCH * kWS + off + 32 * kWS);
      const u32x4 wl0 = *reinterpret_cast<const u32x4*>(lds + 2 * kCH * kWS + off);
      const u32x4 wl1 = *reinterpret_cast<const u32x4*>(lds + 2 * kCH * kWS + off + 32 * kWS);
#ifdef EXP_FS_ONEMFMA       // bounding build: one product instead of six
      acc0 = mfma_bf16(wh0 ^ wm0 ^ wl0, xh[s] ^ xm[s] ^ xl[s], acc0); acc1 = mfma_bf16(wh1 ^ wm1 ^ wl1, xh[s] ^ xm[s] ^ xl[s], acc1);
#else
      acc0 = mfma_bf16(wl0, xh[s], acc0); acc1 = mfma_bf16(wl1, xh[s], acc1);        // smallest terms first
      acc0 = mfma_bf16(wh0, xl[s], acc0); acc1 = mfma_bf16(wh1, xl[s], acc1);
      acc0 = mfma_bf16(wm0, xm[s], acc0); acc1 = mfma_bf16(wm1, xm[s], acc1);
      acc0 = mfma_bf16(wm0, xh[s], acc0); acc1 = mfma_bf16(wm1, xh[s], acc1);
      acc0 = mfma_bf16(wh0, xm[s], acc0); acc1 = mfma_bf16(wh1, xm[s], acc1);
      acc0 = mfma_bf16(wh0, xh[s], acc0); acc1 = mfma_bf16(wh1, xh[s], acc1);
#endif
    }
  };
  auto store_tile = [&](const f32x16& acc, int c0, int ct) {
#ifdef EXP_FS_NOSTORE       // bounding build: no output traffic (one lane's store keeps the arithmetic alive)
    if (lane != 63 || blockIdx.x != 0) return;
#endif
    if (!row_ok) return;
    float* yr = p.y + row * p.ldy + c0;
    const float* tr = p.rowbias ? p.rowbias + (int64_t)p.rowidx[row] * p.ld_rb + c0 : nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = ct * 32 + 8 * g + 4 * h;
      f32x4 v = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      if (p.bias) v = v + *reinterpret_cast<const f32x4*>(p.bias + c0 + col);
      if (tr) v = v + *reinterpret_cast<const f32x4*>(tr + col);
      if (p.gate) {
        const f32x4 gq = *reinterpret_cast<const f32x4*>(p.gate + row * p.ld_gate + c0 + col);
        v = f32x4{v.x * gate_grad(gq.x, p.gate_kind), v.y * gate_grad(gq.y, p.gate_kind),
                  v.z * gate_grad(gq.z, p.gate_kind), v.w * gate_grad(gq.w, p.gate_kind)};
      }
      *reinterpret_cast<f32x4*>(yr + col) = v;
    }
  };

  f32x16 acc[NACC];
  if constexpr (NS == 1) {
    w_fetch(0, 0);
    load_x(0);
    const int n_chunks = M / kCH;
    for (int c = 0; c < n_chunks; ++c) {
      lds_barrier();                                       // every wave has left the previous chunk's reads
      w_commit();
      lds_barrier();
      if (c + 1 < n_chunks) w_fetch((c + 1) * kCH, 0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
      chunk_mma(acc[0], acc[1]);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) store_tile(acc[ct], c * kCH, ct);
    }
  } else {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    // software pipeline over the (slice, chunk) stages: the NEXT slice's fp32 rows and the NEXT stage's W planes are in
    // flight (registers) under this stage's MFMAs -- the slice-by-slice form exposed one load latency per slice and one W
    // staging per stage: 0.68 of its 0.90 ms remained with one MFMA and no stores (tools/build_variant_file.sh fs_*)
#if SEGGER_FS_K384_WAVES == 1
    f32x4 xraw[16];
    auto fetch_x = [&](int slice) {
      const float* xr = p.x + row * p.ldx + slice * kKS + 8 * h;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        xraw[2 * s] = *reinterpret_cast<const f32x4*>(xr + 16 * s);
        xraw[2 * s + 1] = *reinterpret_cast<const f32x4*>(xr + 16 * s + 4);
      }
    };
    fetch_x(0);
    w_fetch(0, 0);
#pragma unroll 1
    for (int slice = 0; slice < NS; ++slice) {
#pragma unroll
      for (int s = 0; s < 8; ++s) split8(xraw[2 * s], xraw[2 * s + 1], xh[s], xm[s], xl[s]);
      if (slice + 1 < NS) fetch_x(slice + 1);
#pragma unroll
      for (int c = 0; c < NACC / 2; ++c) {
#ifdef EXP_FS_NOWSTAGE       // bounding build: W staged once, never again (no W loads, LDS writes or barriers in the loop)
        if (slice == 0 && c == 0) { lds_barrier(); w_commit(); lds_barrier(); }
#else
        lds_barrier();
        w_commit();
        lds_barrier();
        if (c + 1 < NACC / 2) w_fetch((c + 1) * kCH, slice);
        else if (slice + 1 < NS) w_fetch(0, slice + 1);
#endif
        chunk_mma(acc[2 * c], acc[2 * c + 1]);
      }
    }
#else
#pragma unroll 1
    for (int slice = 0; slice < NS; ++slice) {
      load_x(slice);
#pragma unroll
      for (int c = 0; c < NACC / 2; ++c) {
        lds_barrier();
        stage_w(c * kCH, slice);
        lds_barrier();
        chunk_mma(acc[2 * c], acc[2 * c + 1]);
      }
    }
#endif
#pragma unroll
    for (int a = 0; a < NACC; ++a) store_tile(acc[a], (a / 2) * kCH, a & 1);
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[M, K] = dY[n, M]^T X[n, K], db = sum_n dY for fp32 operands on the bf16 matrix pipe: both operands split three ways
// as above, the product as its six leading partial products.  The exact-fp32 kernel (csrc/linear_f32.hip,
// v_mfma_f32_32x32x2_f32) is MFMA-bound at 0.94 ms for 1M x (384, 128); the six bf16 products are 0.375 of that matrix time.
// Structure (as csrc/linear_wgrad.hip, without its LDS-DMA -- the fp32 rows have to pass through the VALU to be split):
// a persistent workgroup owns a slab of rows and keeps the whole [M, K] accumulator in its waves' registers; 16 rows per
// stage: every thread loads its 16-byte pieces of the stage's dY / X rows (the NEXT stage's loads are in flight under
// this stage's MFMAs), splits them and writes three bf16 planes per matrix into one of two LDS buffers (row stride
// width * 2 + 64 bytes: the transposing reads are conflict-free); the MFMA operands -- the row index is the k dimension
// of both -- come out of LDS transposed by ds_read_b64_tr_b16.  One barrier per stage.  Partial sums per workgroup in the
// layout of the other weight-gradient kernels (summed in slab order by reduce_partials: deterministic).
typedef short s16x4s __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2s lds_read_tr_s(const unsigned char* p) {
  const s16x4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4s*)(const_cast<unsigned char*>(p)));
  return __builtin_bit_cast(u32x2s, v);
}

struct WgSplitParams {
  const float* dy; int64_t ld_dy;
  const float* x; int64_t ld_x;
  int64_t n_rows, n_stages, stages_per_block;
  float* partial;            // [gridDim.x][M * K + M]
};

// 4 consecutive floats -> 4 bf16 of each plane
__device__ __forceinline__ void split4(const f32x4 v, u32x2s& hi, u32x2s& mid, u32x2s& lo) {
#ifdef EXP_FS_NOSPLIT
  hi = u32x2s{__float_as_uint(v.x), __float_as_uint(v.y)}; mid = u32x2s{__float_as_uint(v.z), __float_as_uint(v.w)};
  lo = u32x2s{__float_as_uint(v.x) + 1u, __float_as_uint(v.w)};
  return;
#endif
  const uint32_t h0 = Vec8<bf16_t>::pack(v.x, v.y), h1 = Vec8<bf16_t>::pack(v.z, v.w);
  float a, b, c, d;
  Vec8<bf16_t>::unpack2(h0, a, b); Vec8<bf16_t>::unpack2(h1, c, d);
  const float r0 = v.x - a, r1 = v.y - b, r2 = v.z - c, r3 = v.w - d;
  const uint32_t m0 = Vec8<bf16_t>::pack(r0, r1), m1 = Vec8<bf16_t>::pack(r2, r3);
  Vec8<bf16_t>::unpack2(m0, a, b); Vec8<bf16_t>::unpack2(m1, c, d);
  hi = u32x2s{h0, h1}; mid = u32x2s{m0, m1};
  lo = u32x2s{Vec8<bf16_t>::pack(r0 - a, r1 - b), Vec8<bf16_t>::pack(r2 - c, r3 - d)};
}

constexpr int kWgRows = 16;                    // rows per stage = one 32x32x16 k-step
template <int N, typename F>
__device__ __forceinline__ void static_for_wgs(F&& f) {
  if constexpr (N > 0) {
    static_for_wgs<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int M, int K, int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_f32_split_kernel(WgSplitParams p) {
  constexpr int NT = NW * 64;
  constexpr int WM = NW == 8 ? 4 : 2, WK = 2;
  constexpr int TM = M / 32, TK = K / 32;
  static_assert(TM % WM == 0 && TK % WK == 0, "tile grid does not split over the waves");
  constexpr int MT = TM / WM, KT = TK / WK;
  constexpr int SY = M * 2 + 64, SX = K * 2 + 64;                 // bytes per row of a plane
  constexpr int PY = kWgRows * SY, PX = kWgRows * SX;             // one plane of a stage
  constexpr int BUF = 3 * PY + 3 * PX;
  // 16-byte pieces of a stage per thread: first its dY pieces, then its X pieces (which matrix piece j belongs to is a
  // compile-time fact: no per-lane pointer select, no branch around the bias sums)
  static_assert((kWgRows * M / 4) % NT == 0 && (kWgRows * K / 4) % NT == 0, "pieces do not split evenly over the workgroup");
  constexpr int NPY = kWgRows * M / 4 / NT, NPX = kWgRows * K / 4 / NT;
  constexpr int PIECES = NPY + NPX;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wk = wave / WM;

  const int64_t s_beg = (int64_t)blockIdx.x * p.stages_per_block;
  int64_t s_end = s_beg + p.stages_per_block;
  if (s_end > p.n_stages) s_end = p.n_stages;

  // this thread's pieces of a stage: fixed (row, column) for the whole kernel
  int prow[PIECES], pcol[PIECES], pdst[PIECES];
  bool pis_y[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    pis_y[j] = j < NPY;
    const int q = tid + (pis_y[j] ? j : j - NPY) * NT, ppr = (pis_y[j] ? M : K) / 4;
    prow[j] = q / ppr;
    pcol[j] = 4 * (q % ppr);
    pdst[j] = pis_y[j] ? prow[j] * SY + pcol[j] * 2 : 3 * PY + prow[j] * SX + pcol[j] * 2;
  }
#ifndef SEGGER_WGS_AHEAD
#define SEGGER_WGS_AHEAD 1                     // stages of global loads in flight (register slots): 1M x (384, 128) on one
                                               // MI355X: 1 -> 0.64 ms, 2 -> 0.99, 3 -> 0.76 (exact fp32: 1.19; one MFMA and no
                                               // split: 0.41 = the 2 GB of fp32 operands at the achievable HBM rate)
#endif
  constexpr int AH = SEGGER_WGS_AHEAD;
  f32x4 ring[AH][PIECES], dbp[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) dbp[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](auto slot_c, int64_t s) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      int64_t row = s * kWgRows + prow[j];
      const bool ok = row < p.n_rows && s < s_end;
      if (!ok) row = p.n_rows - 1;                                // clamp: loaded, zeroed below
      const float* src = pis_y[j] ? p.dy + row * p.ld_dy + pcol[j] : p.x + row * p.ld_x + pcol[j];
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      ring[SL][j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  // transposing-read addresses (csrc/linear_wgrad.hip): lane (g, i16) reads 4 rows x 16 columns blocks
  const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const int row_a = 8 * (g >> 1) + tq, col_a = 16 * (g & 1) + 4 * tp;
  const int off_y = row_a * SY + col_a * 2 + wm * MT * 64;
  const int off_x = 3 * PY + row_a * SX + col_a * 2 + wk * KT * 64;

  f32x16 acc[MT][KT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // one stage: split the ring slot's rows into the LDS buffer, refill the slot with the stage AH ahead (in flight under
  // this and the next stages' MFMAs), barrier, MFMAs
  auto stage = [&](auto slot_c, int64_t s) {
    constexpr int SL = decltype(slot_c)::value;
    unsigned char* buf = lds + ((s - s_beg) & 1) * BUF;
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      u32x2s h, m, l;
#ifdef EXP_WGS_NOSPLIT      // bounding build: no VALU split (planes = raw words)
      h = u32x2s{__float_as_uint(ring[SL][j].x), __float_as_uint(ring[SL][j].y)};
      m = u32x2s{__float_as_uint(ring[SL][j].z), __float_as_uint(ring[SL][j].w)}; l = h;
#else
      split4(ring[SL][j], h, m, l);
#endif
      const int plane = pis_y[j] ? PY : PX;
      *reinterpret_cast<u32x2s*>(buf + pdst[j]) = h;
      *reinterpret_cast<u32x2s*>(buf + pdst[j] + plane) = m;
      *reinterpret_cast<u32x2s*>(buf + pdst[j] + 2 * plane) = l;
      if (pis_y[j]) dbp[j] = dbp[j] + ring[SL][j];
    }
    fetch(slot_c, s + AH);
    lds_barrier();                                             // (also: every wave has left the reads of stage s - 2)
    u32x4 fa[3][MT], fb[3][KT];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const u32x2s lo = lds_read_tr_s(buf + q * PY + off_y + a * 64);
        const u32x2s hi = lds_read_tr_s(buf + q * PY + off_y + a * 64 + 4 * SY);
        fa[q][a] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
#pragma unroll
      for (int b = 0; b < KT; ++b) {
        const u32x2s lo = lds_read_tr_s(buf + q * PX + off_x + b * 64);
        const u32x2s hi = lds_read_tr_s(buf + q * PX + off_x + b * 64 + 4 * SX);
        fb[q][b] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < KT; ++b) {
        f32x16 c = acc[a][b];
#ifdef EXP_WGS_ONEMFMA      // bounding build: one product instead of six
        c = mfma_bf16(fa[0][a] ^ fa[1][a] ^ fa[2][a], fb[0][b] ^ fb[1][b] ^ fb[2][b], c);
#else
        c = mfma_bf16(fa[2][a], fb[0][b], c);                    // smallest terms first
        c = mfma_bf16(fa[0][a], fb[2][b], c);
        c = mfma_bf16(fa[1][a], fb[1][b], c);
        c = mfma_bf16(fa[1][a], fb[0][b], c);
        c = mfma_bf16(fa[0][a], fb[1][b], c);
        c = mfma_bf16(fa[0][a], fb[0][b], c);
#endif
        acc[a][b] = c;
      }
  };
#ifndef SEGGER_WGS_PIPE
#define SEGGER_WGS_PIPE 1                      // stage s's MFMAs with stage s + 1's split between them (below)
#endif
#if SEGGER_WGS_PIPE && SEGGER_WGS_AHEAD == 1
  // The form above runs a VALU phase (the split: ~170 instructions) and an MFMA phase (36 x 8 passes) per stage one after
  // the other -- and the workgroup's barrier keeps both waves of a SIMD in the SAME phase, so nothing overlaps them
  // (measured 0.55 ms = their sum, against 0.33 for the MFMAs alone at the clock this data allows).  Software pipeline:
  // iteration s reads stage s's fragments (written to buffer s & 1 during iteration s - 1) and issues its MFMAs with the
  // split of stage s + 1's rows -- into the other buffer, whose readers all passed this iteration's barrier -- between
  // them, a few VALU per MFMA (one wave issues in order: only what sits BETWEEN two MFMAs runs in the first one's shadow),
  // each ring register refilled with stage s + 2's piece right after its split.  Still one barrier per stage.
  constexpr int NCH = PIECES * 5;              // side chunks of an iteration: per piece 3 of arithmetic, 1 of LDS writes, 1 load
  constexpr int NMF = MT * KT * 6;
  pk2 sva[PIECES], svb[PIECES], sra[PIECES], srb[PIECES];
  uint32_t sh[PIECES][2], sm[PIECES][2], sl[PIECES][2];
  auto up = [](uint32_t w) { float a, b; Vec8<bf16_t>::unpack2(w, a, b); return pk2{a, b}; };
  // the block's slab of each matrix as a raw buffer (rows past the matrix: out of range = zeros; stages past the slab: the
  // out-of-range sentinel): 32-bit offsets inside the slab, no clamp and no select on loaded data
  const int64_t slab_row = s_beg * kWgRows;
  auto slab = [&](const float* base, int64_t ld) {
    const int64_t left = p.n_rows > slab_row ? (p.n_rows - slab_row) * ld * 4 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + slab_row * ld), 0, (int)(uint32_t)(left < 0xfffff000LL ? left : 0xfffff000LL),
                                             0x00020000);
  };
  const __amdgpu_buffer_rsrc_t yb = slab(p.dy, p.ld_dy), xb = slab(p.x, p.ld_x);
  uint32_t poff[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) poff[j] = (uint32_t)(prow[j] * (pis_y[j] ? p.ld_dy : p.ld_x) + pcol[j]) * 4u;
  const uint32_t sy4 = (uint32_t)(kWgRows * p.ld_dy * 4), sx4 = (uint32_t)(kWgRows * p.ld_x * 4);
  auto fetch_piece = [&](int j, int64_t st) {                   // st: absolute stage
    const uint32_t sl_ = (uint32_t)(st - s_beg);
    const uint32_t off = st < s_end ? poff[j] + sl_ * (pis_y[j] ? sy4 : sx4) : 0xfffff000u;
    ring[0][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pis_y[j] ? yb : xb, off, 0, 0));
  };
  auto side = [&](auto c_c, unsigned char* nbuf, int64_t st) {  // st: the stage whose MFMAs run; the ring holds stage st + 1
    constexpr int c = decltype(c_c)::value;
    if constexpr (c < NCH) {
      constexpr int j = c / 5, ph = c % 5;
      if constexpr (ph == 0) {
        const f32x4 v = ring[0][j];
        sva[j] = pk2{v.x, v.y}; svb[j] = pk2{v.z, v.w};
        if (pis_y[j]) dbp[j] = dbp[j] + v;
        sh[j][0] = Vec8<bf16_t>::pack(v.x, v.y); sh[j][1] = Vec8<bf16_t>::pack(v.z, v.w);
      } else if constexpr (ph == 1) {
        sra[j] = sva[j] - up(sh[j][0]); srb[j] = svb[j] - up(sh[j][1]);
        sm[j][0] = Vec8<bf16_t>::pack(sra[j].x, sra[j].y); sm[j][1] = Vec8<bf16_t>::pack(srb[j].x, srb[j].y);
      } else if constexpr (ph == 2) {
        const pk2 la = sra[j] - up(sm[j][0]), lb = srb[j] - up(sm[j][1]);
        sl[j][0] = Vec8<bf16_t>::pack(la.x, la.y); sl[j][1] = Vec8<bf16_t>::pack(lb.x, lb.y);
      } else if constexpr (ph == 3) {
        const int plane = pis_y[j] ? PY : PX;
        *reinterpret_cast<u32x2s*>(nbuf + pdst[j]) = u32x2s{sh[j][0], sh[j][1]};
        *reinterpret_cast<u32x2s*>(nbuf + pdst[j] + plane) = u32x2s{sm[j][0], sm[j][1]};
        *reinterpret_cast<u32x2s*>(nbuf + pdst[j] + 2 * plane) = u32x2s{sl[j][0], sl[j][1]};
      } else {
        fetch_piece(j, st + 2);
      }
    }
  };
  // prologue: stage s_beg's planes into buffer 0, stage s_beg + 1's rows into the ring
#pragma unroll
  for (int j = 0; j < PIECES; ++j) fetch_piece(j, s_beg);
  static_for_wgs<NCH>([&](auto c_c) {
    if constexpr (decltype(c_c)::value % 5 != 4) side(c_c, lds, s_beg - 1);
  });
#pragma unroll
  for (int j = 0; j < PIECES; ++j) { fetch_piece(j, s_beg + 1); __builtin_amdgcn_sched_barrier(0); }
#ifdef EXP_WGS_STAMPS
  long long st_bar = 0, st_tot = 0, st_n = 0, st_prev = clock64();
#endif
#pragma unroll 1
  for (int64_t s = s_beg; s < s_end; ++s) {
    unsigned char* buf = lds + ((s - s_beg) & 1) * BUF;
    unsigned char* nbuf = lds + (((s - s_beg) & 1) ^ 1) * BUF;
#ifdef EXP_WGS_STAMPS
    const long long wc0 = clock64();
#endif
    lds_barrier();             // stage s's planes are complete; every wave has left the other buffer (stage s - 1)
#ifdef EXP_WGS_STAMPS
    const long long wc1 = clock64();
    st_bar += wc1 - wc0; st_tot += wc1 - st_prev; st_prev = wc1; ++st_n;
#endif
    // fragments in the order the tiles need them: the first MFMA waits for one row and one column block, the rest arrive
    // under the MFMAs in front of their first use (both waves of a SIMD leave the barrier together: with all 30 reads first
    // the matrix pipe idles through them)
    u32x4 fa[3][MT], fb[3][KT];
    auto read_a = [&](int a) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const u32x2s lo = lds_read_tr_s(buf + q * PY + off_y + a * 64);
        const u32x2s hi = lds_read_tr_s(buf + q * PY + off_y + a * 64 + 4 * SY);
        fa[q][a] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
    };
    auto read_b = [&](int b) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const u32x2s lo = lds_read_tr_s(buf + q * PX + off_x + b * 64);
        const u32x2s hi = lds_read_tr_s(buf + q * PX + off_x + b * 64 + 4 * SX);
        fb[q][b] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
    };
    read_a(0); read_b(0);
    __builtin_amdgcn_sched_barrier(0);
    static_for_wgs<NMF>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      constexpr int tile = i / 6, pr = i % 6, a = tile / KT, b = tile % KT;
      // column block b + 1 behind the first MFMA of row block 0; row block a + 1 behind the first MFMA of row block a's second tile
      if constexpr (a == 0 && pr == 1 && b + 1 < KT) read_b(b + 1);
      if constexpr (pr == 1 && b == (KT > 1 ? 1 : 0) && a + 1 < MT) read_a(a + 1);
      constexpr int qa = pr == 0 ? 2 : (pr == 2 || pr == 3) ? 1 : 0;       // smallest terms first (as above)
      constexpr int qb = pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
      acc[a][b] = mfma_bf16(fa[qa][a], fb[qb][b], acc[a][b]);
      // chunks [i * NCH / NMF, (i + 1) * NCH / NMF) of the side work behind this MFMA
      static_for_wgs<(i + 1) * NCH / NMF - i * NCH / NMF>([&](auto d_c) {
        side(std::integral_constant<int, i * NCH / NMF + decltype(d_c)::value>{}, nbuf, s);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
#ifdef EXP_WGS_STAMPS
  if (blockIdx.x == 5 && lane == 0 && st_n > 0)
    printf("wave %d: %lld stages, per stage %lld clocks, of which at the barrier %lld\n", wave, st_n, st_tot / st_n, st_bar / st_n);
#endif
#else
  static_for_wgs<AH>([&](auto d) { fetch(d, s_beg + decltype(d)::value); });
  int64_t s = s_beg;
#pragma unroll 1
  for (; s + AH <= s_end; s += AH)
    static_for_wgs<AH>([&](auto d) { stage(d, s + decltype(d)::value); });
  static_for_wgs<AH>([&](auto d) { if (s + decltype(d)::value < s_end) stage(d, s + decltype(d)::value); });
#endif

  // acc tile (a, b) element e of lane l is dW[m][k]: m = 32 (wm MT + a) + (e & 3) + 8 (e >> 2) + 4 (l >> 5), k = 32 (wk KT + b) + (l & 31)
  float* out = p.partial + (int64_t)blockIdx.x * ((int64_t)M * K + M);
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = 32 * (wm * MT + a) + (e & 3) + 8 * (e >> 2) + 4 * hh;
        out[m * K + 32 * (wk * KT + b) + r] = acc[a][b][e];
      }
  // db: every thread holds the sums of its dY pieces' columns over its row of every stage; the 16 rows meet in LDS
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);                    // [16][M]
#pragma unroll
  for (int j = 0; j < PIECES; ++j)
    if (pis_y[j]) *reinterpret_cast<f32x4*>(red + prow[j] * M + pcol[j]) = dbp[j];
  __syncthreads();
  for (int m = tid; m < M; m += NT) {
    float sum = 0.f;
#pragma unroll
    for (int rr = 0; rr < kWgRows; ++rr) sum += red[rr * M + m];
    out[M * K + m] = sum;
  }
}

constexpr int split_waves(int m, int k) { return (m / 32) * (k / 32) >= 32 ? 8 : 4; }
constexpr int64_t kSplitMinStages = 32;        // 512 rows per workgroup at least

}  // namespace

bool wgrad_f32_split_shape_ok(int m, int k) {
  // (instantiated for (64, 64) and (64, 128) as well, where the exact kernel is faster -- 0.14 / 0.23 vs 0.20 / 0.26 ms: few
  //  MFMAs per staged row)
  return (m == 384 && k == 128) || (m == 128 && (k == 128 || k == 256)) || (m == 64 && k == 256);
}

int64_t wgrad_f32_split_grid(int64_t n_rows, int m, int k) {
  const int64_t stages = (n_rows + kWgRows - 1) / kWgRows;
  const int64_t want = (stages + kSplitMinStages - 1) / kSplitMinStages;
  const int64_t cap = split_waves(m, k) == 8 ? device_cu_count() : 2 * (int64_t)device_cu_count();
  const int64_t c = cap > 0 ? cap : 256;
  return want < c ? (want < 1 ? 1 : want) : c;
}

int wgrad_f32_split_launch(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t n_rows, int m_out, int k_in,
                           float* partial, int64_t* n_slabs, hipStream_t stream) {
  const int64_t grid = wgrad_f32_split_grid(n_rows, m_out, k_in);
  WgSplitParams p{dy, ld_dy, x, ld_x, n_rows, (n_rows + kWgRows - 1) / kWgRows, 0, partial};
  p.stages_per_block = (p.n_stages + grid - 1) / grid;
  *n_slabs = grid;
  // (the kernel addresses a workgroup's slab of rows with 32-bit byte offsets)
  SEGGER_REQUIRE((p.stages_per_block + 2) * kWgRows * (ld_dy > ld_x ? ld_dy : ld_x) * 4 < 0xfffff000LL,
                 "segger_linear_wgrad_f32_split: a workgroup's row slab exceeds 4 GiB");
#define CASE(MM, KK) if (m_out == MM && k_in == KK) { \
    hipLaunchKernelGGL((wgrad_f32_split_kernel<MM, KK, split_waves(MM, KK)>), dim3((unsigned)grid), dim3(split_waves(MM, KK) * 64), 0, stream, p); \
    SEGGER_LAUNCH_CHECK("wgrad_f32_split_kernel"); return SEGGER_OK; }
  CASE(384, 128) CASE(128, 128) CASE(128, 256) CASE(64, 64) CASE(64, 128) CASE(64, 256)
#undef CASE
  set_error("segger_linear_wgrad_f32_split: m_out=%d k_in=%d not supported", m_out, k_in);
  return SEGGER_EUNSUPPORTED;
}

}  // namespace segger

using namespace segger;

namespace segger { namespace {
// planes[q][o] (q = hi, mid, lo) of w [rows, cols] fp32, o running over the matrix in its own order or, `transpose`, over w^T
__global__ __launch_bounds__(256) void f32_split_planes_kernel(const float* __restrict__ w, int rows, int cols, int transpose,
                                                              bf16_t* __restrict__ out) {
  const int64_t n = (int64_t)rows * cols, o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const float v = transpose ? w[(o % rows) * cols + o / rows] : w[o];
  const uint32_t h = Vec8<bf16_t>::pack(v, 0.f);
  float hf, mf, z;
  Vec8<bf16_t>::unpack2(h, hf, z);
  const float r = v - hf;
  const uint32_t m = Vec8<bf16_t>::pack(r, 0.f);
  Vec8<bf16_t>::unpack2(m, mf, z);
  const uint32_t l = Vec8<bf16_t>::pack(r - mf, 0.f);
  out[o].v = (uint16_t)(h & 0xffffu); out[n + o].v = (uint16_t)(m & 0xffffu); out[2 * n + o].v = (uint16_t)(l & 0xffffu);
}

struct PlanesJobs { segger_planes_job job[SEGGER_PLANES_MAX_JOBS]; };
// the same for several matrices: blockIdx.y = job, blockIdx.x over the largest job's elements (smaller jobs' extra blocks exit)
__global__ __launch_bounds__(256) void f32_split_planes_many_kernel(PlanesJobs all) {
  const segger_planes_job& j = all.job[blockIdx.y];
  const int rows = j.rows, cols = j.cols;
  const int64_t n = (int64_t)rows * cols, o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const float* __restrict__ w = j.w;
  bf16_t* __restrict__ out = static_cast<bf16_t*>(j.planes);
  const float v = j.transpose ? w[(o % rows) * cols + o / rows] : w[o];
  const uint32_t h = Vec8<bf16_t>::pack(v, 0.f);
  float hf, mf, z;
  Vec8<bf16_t>::unpack2(h, hf, z);
  const float r = v - hf;
  const uint32_t m = Vec8<bf16_t>::pack(r, 0.f);
  Vec8<bf16_t>::unpack2(m, mf, z);
  const uint32_t l = Vec8<bf16_t>::pack(r - mf, 0.f);
  out[o].v = (uint16_t)(h & 0xffffu); out[n + o].v = (uint16_t)(m & 0xffffu); out[2 * n + o].v = (uint16_t)(l & 0xffffu);
}

// ------------------------------------------------------------------------------- W resident in registers (K * M = 49152)
// The two big shapes -- 128 -> 384 (stacked lin_l | lin_r | lin_l) and its data gradient 384 -> 128 -- spent their time in
// SERIES: the bounding builds of linear_f32_split_kernel<384> (tools/ab_fs384.sh; 0.872 ms at 1M rows) give back 0.19 ms
// without five of the six products, 0.14 without the output stores, 0.12 without the W staging, 0.065 without the split --
// with one wave per SIMD nothing runs under anything else.  Here the roles are swapped.  A persistent workgroup (one per CU,
// 4 waves) keeps ALL of W in registers for the whole launch: wave w owns M / 4 output columns, 3 planes x M/4 x K bf16 =
// 288 registers per lane as MFMA A-fragments.  The ROWS stream through LDS: a tile of 32 rows is loaded as fp32 by all
// 256 threads (coalesced 16-byte pieces, one tile AHEAD in registers), split into its three bf16 planes and written to one
// of two LDS buffers; every wave reads the tile's B-fragments from there (conflict-free ds_read_b128, row stride 2 K + 16).
// One LDS-only barrier per tile and no W traffic in the loop.
// The memory stream is what decides the speed of a kernel with ONE wave per SIMD -- nothing else runs while it waits, and
// vmcnt counts loads and stores in one in-order queue.  So: per tile the loop issues, in this order, the row loads of tile
// t + 2 (piece by piece, each right after its register was split into the planes of tile t + 1: 48 KB per CU in flight all
// the time, not just between a block of loads and its use), the output stores of tile t - 1 (kept in 16 registers per column tile; its gate values were loaded a tile earlier)
// and the gate loads of tile t; the wait before the split of tile t + 1's rows is then vmcnt(stores + gate loads issued
// after them): the loads have had a whole tile to arrive and no store is ever waited for.  All of it branch-free (buffer
// loads / stores: rows past the end are out of range, dropped by the memory unit) so the counter arithmetic stays exact.
template <int K, int M, int GATE, bool BIAS>          // GATE: 0 none, 1 GELU, 2 SiLU (p.gate_kind at compile time: no branch in the loop)
__global__ __launch_bounds__(256, 1) void linear_f32_split_wres_kernel(SplitParams p, int n_tiles) {
  static_assert(K * M == 384 * 128 && K % 128 == 0 && M % 128 == 0, "W must fill 288 registers per lane");
  constexpr int NKS = K / 16;                  // k-steps of a tile: 24 / 8
  constexpr int NCT = M / 128;                 // 32-column tiles per wave: 1 / 3
  constexpr int NPAR = NCT == 1 ? 2 : 1;       // independent accumulator chains per column tile (NCT * NPAR >= 2)
  constexpr int XS = K * 2 + 16;               // LDS row stride of a plane (bytes)
  constexpr int PLANE = 32 * XS, BUF = 3 * PLANE;
  constexpr int CPL = K / 128;                 // 16-byte pieces per lane and row (32 lanes per row): 3 / 1
  constexpr int NP = 4 * CPL;                  // pieces per thread and tile (rows rg, rg + 8, rg + 16, rg + 24): 12 / 4
  constexpr int HEAD = NKS / 3;                // k-steps before the split starts
  static_assert(HEAD + NP <= NKS, "the split must fit behind the head");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  __shared__ __attribute__((aligned(16))) float lds_bias[BIAS ? M : 4];     // (no global load in the store path: see below)
  // K = 128 leaves LDS for the finished tile: it waits for its stores THERE, in row-major order, and leaves as full 128-byte
  // lines (8 lanes x 16 bytes per row and instruction).  From the accumulator layout a store instruction touches 32 rows
  // with 32 bytes each, and the per-CU write path takes a store by the line: in-kernel stamps showed ~210 cycles of issue
  // per such store, 12 of them per wave and tile -- as much as the tile's MFMAs.
  constexpr bool LDS_OUT = K == 128;
  constexpr int OS = 36;                                                    // floats per row of an output tile in LDS (32 + pad)
  __shared__ __attribute__((aligned(16))) float lds_out[LDS_OUT ? 4 * NCT * 32 * OS : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int l32 = tid & 31, rg = tid >> 5;
  const int G = gridDim.x;
  int t = blockIdx.x;
  if (t >= n_tiles) return;
  if constexpr (BIAS) {
    for (int c = tid; c < M; c += 256) lds_bias[c] = p.bias[c];            // (ordered before its first read by the loop's barrier)
  }

  u32x4 wf[3][NCT][NKS];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int s = 0; s < NKS; ++s)
        wf[q][ct][s] = *reinterpret_cast<const u32x4*>(p.w3 + ((int64_t)q * M + (wave * NCT + ct) * 32 + r) * K + 16 * s + 8 * h);
#ifndef EXP_FS_WF_VGPR
  // two of the three planes pinned to accumulation registers (the MFMA reads its A operand from either file): no
  // v_accvgpr_read copies in front of the MFMAs
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int s = 0; s < NKS; ++s) asm volatile("" : "+a"(wf[q][ct][s]));
#endif

  // x, y, gate as raw buffers (the host checked that they are below 2 GB): 32-bit offsets, out-of-range lanes dropped --
  // kNoRow + any column offset stays out of range
  constexpr uint32_t kNoRow = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(uint32_t)(p.n_rows * p.ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(uint32_t)(p.n_rows * p.ldy * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t gb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gate), 0, GATE ? (int)(uint32_t)(p.n_rows * p.ld_gate * 4) : 0, 0x00020000);
  const uint32_t ldx4 = (uint32_t)p.ldx * 4u, ldy4 = (uint32_t)p.ldy * 4u, ldg4 = (uint32_t)p.ld_gate * 4u;
  const uint32_t n_rows = (uint32_t)p.n_rows;

  f32x4 raw[NP];
  auto fetch_piece = [&](int tile, int j) {    // (rows past the end: zeros)
    const uint32_t row = (uint32_t)tile * 32u + rg + 8 * (j / CPL);
    const uint32_t off = row < n_rows ? row * ldx4 + l32 * 16u : kNoRow;
    raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xb, off + (j % CPL) * 512, 0, 0));
  };
  auto fetch = [&](int tile) {                  // in piece order, pinned: the loop's counted waits (vmcnt(NP - 1 + later
#pragma unroll                                  // stores / gate loads) before piece 0's split, ...) are derived from the
    for (int j = 0; j < NP; ++j) {               // order of issue on BOTH ways into the loop -- a prologue that loads piece 0
      fetch_piece(tile, j);                      // last (the scheduler did) turns every one of them into vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  unsigned char* const cbase = lds + rg * XS + l32 * 8;
  auto commit_piece = [&](int buf_off, int j) {
    u32x2s hi, mid, lo;
    split4(raw[j], hi, mid, lo);
    unsigned char* d = cbase + buf_off + (j / CPL) * 8 * XS + (j % CPL) * 256;
    *reinterpret_cast<u32x2s*>(d) = hi;
    *reinterpret_cast<u32x2s*>(d + PLANE) = mid;
    *reinterpret_cast<u32x2s*>(d + 2 * PLANE) = lo;
  };
  // the finished tile waiting for its store: accumulators, byte offsets of this lane's row in y / gate (~0: no row), gate
  f32x16 pend[NCT];
  uint32_t pend_y = kNoRow;
  f32x4 gq[GATE ? NCT * 4 : 1];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int e = 0; e < 16; ++e) pend[ct][e] = 0.f;
  if constexpr (GATE) {
#pragma unroll
    for (int g = 0; g < NCT * 4; ++g) gq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const uint32_t col0 = (uint32_t)(wave * NCT * 32 + 4 * h) * 4u;       // byte offset of this lane's first column
  auto gate_apply = [&](int e) {                // pend *= act'(gate), one value (the last tile; the loop does it in stages)
    if constexpr (GATE != 0) {
      static_assert(NCT == 1 && 2 * HEAD == 16, "gate: the 384 -> 128 shape, two values per HEAD k-step");
      pend[0][e] *= gate_grad(gq[e / 4][e % 4], GATE);
    }
  };
  auto store_piece = [&](int q) {                // 16 bytes of the stored tile: column tile q / 4, group q % 4
    const int ct = q / 4, g = q % 4;
    const f32x4 v = f32x4{pend[ct][4 * g], pend[ct][4 * g + 1], pend[ct][4 * g + 2], pend[ct][4 * g + 3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yb, pend_y + (ct * 32 + 8 * g) * 4, 0, 0);
  };
  // LDS_OUT: piece q = rows 8 (q % 4) .. + 8 of column tile q / 4, read back row-major (lane: row lane / 8, 16 bytes lane % 8)
  float* const out_w = lds_out + (LDS_OUT ? wave * NCT * 32 * OS : 0);
  uint32_t pend_row0 = kNoRow;                   // first row of the tile waiting in LDS (kNoRow: none)
  f32x4 ov[3];                                   // (read two slots ahead of its store: an LDS read under load outlasts one slot)
  auto out_read = [&](int q) {
    ov[q % 3] = *reinterpret_cast<const f32x4*>(out_w + ((q / 4) * 32 + 8 * (q % 4) + (lane >> 3)) * OS + 4 * (lane & 7));
  };
  // (LDS_OUT: the bias joins at the store, on the row-major piece -- the epilogue then moves the accumulators to LDS as they
  //  are, no copy to arithmetic registers and no add between the tile's last MFMA and the next tile's first)
  f32x4 bias_rm[LDS_OUT && BIAS ? NCT : 1];
  if constexpr (LDS_OUT && BIAS) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bias_rm[ct] = *reinterpret_cast<const f32x4*>(p.bias + (wave * NCT + ct) * 32 + 4 * (lane & 7));
  }
  auto out_store = [&](int q) {
    if constexpr (LDS_OUT && BIAS) ov[q % 3] = ov[q % 3] + bias_rm[q / 4];
    const uint32_t row = pend_row0 + 8 * (q % 4) + (lane >> 3);
    const uint32_t off = (pend_row0 != kNoRow && row < n_rows) ? row * ldy4 + ((wave * NCT + q / 4) * 32 + 4 * (lane & 7)) * 4u : kNoRow;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov[q % 3]), yb, off, 0, 0);
  };
  auto store_pend = [&]() {
    if constexpr (LDS_OUT) {
#pragma unroll
      for (int q = 0; q < NCT * 4; ++q) { out_read(q); out_store(q); }
    } else {
#pragma unroll
      for (int q = 0; q < NCT * 4; ++q) store_piece(q);
    }
  };
  auto load_gate = [&](uint32_t off) {
    if constexpr (GATE) {
#pragma unroll
      for (int g = 0; g < NCT * 4; ++g)
        gq[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gb, off + ((g / 4) * 32 + 8 * (g % 4)) * 4, 0, 0));
    }
  };

  fetch(t);
#pragma unroll
  for (int j = 0; j < NP; ++j) commit_piece(0, j);
  load_gate(kNoRow);         // (no row: zeros / dropped) -- the queue as every later iteration leaves it: gate loads, row
  fetch(t + G);              // loads, stores; the compiler's counts at the loop head are then the steady-state ones
  store_pend();
  int b = 0;
#ifdef EXP_FS_STAMPS
  long long st_bar = 0, st_head = 0, st_commit = 0, st_tail = 0, st_n = 0, st_t0 = clock64();
#endif
  for (; t < n_tiles; t += G, b ^= 1) {
#ifdef EXP_FS_STAMPS
    const long long c0 = clock64();
#endif
    lds_barrier();           // tile t's planes are in buffer b; every wave has left buffer b ^ 1 (tile t - G)
    const unsigned char* cur = lds + b * BUF + r * XS + 16 * h;
    const int nxt = (b ^ 1) * BUF;
    const uint32_t row = (uint32_t)t * 32u + r;
    const uint32_t cur_y = row < n_rows ? row * ldy4 + col0 : kNoRow;
    const uint32_t cur_g = row < n_rows ? row * ldg4 + col0 : kNoRow;
    f32x16 acc[NCT][NPAR];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int q = 0; q < NPAR; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][q][e] = 0.f;
    u32x4 xf[2][3];                                // B-fragments (hi, mid, lo) of this and the next k-step
#pragma unroll
    for (int q = 0; q < 3; ++q) xf[0][q] = *reinterpret_cast<const u32x4*>(cur + q * PLANE);
#ifdef EXP_FS_STAMPS
    long long c1 = 0, c2 = 0, c3 = 0;
#endif
    // In-order issue: an MFMA behind an MFMA waits out the first one's 8 passes, with everything behind it.  The side work
    // only runs in the shadow of the matrix pipe if it sits BETWEEN the MFMAs (about 4 VALU fit free, tools/micro/
    // mfma_shadow.hip) -- and at this register count the machine scheduler keeps the source order (it reverts its own
    // schedule when that raises the pressure; sched_group_barrier pipelines were ignored), so the source IS the schedule.
    // A tile is NKS x 6 x NCT MFMA slots.  Behind slot i:
    //   * gated form, k-steps below HEAD: one stage of the stored tile's gate factors (two values per k-step, register pairs)
    //   * from slot S0 on (0 un-gated, HEAD x 6 gated), spread evenly over SPAN slots: the NP pieces of the next tile's rows
    //     -> planes, eight chunks each (six of the split's arithmetic, the LDS writes, the refill of the piece's register
    //     with the rows of the tile after that: a piece is in flight for a whole tile); before the first chunk this tile's
    //     gate values on their way, behind the last one the previous tile's stores.
    // (NCT = 3: a slot is three MFMAs, there the pieces stay on k-steps HEAD .. HEAD + NP - 1 -- spread over the whole tile the
    //  forward measured 5 % slower)
    constexpr int S0 = (GATE != 0 || NCT > 1) ? HEAD * 6 : 0, SPAN = NCT > 1 ? NP * 6 : NKS * 6 - S0, CPP = 8, NCHK = NP * CPP;
    static_assert(S0 + SPAN <= NKS * 6, "the split must end with the tile");
    // NCT = 3 stores twelve 1-KB pieces per wave and tile: issued in one burst behind the last chunk they held the wave for
    // thousands of cycles (in-kernel stamps: 3.0 - 6.8 k cycles for the four k-steps around them, 2.3 k of MFMAs) -- one piece
    // behind each of the slots that follow the split
    constexpr bool SPREAD_STORES = NCT > 1 && S0 + SPAN + NCT * 4 <= NKS * 6;
    pk2 gx, gsa, gpdf, gq2, ghalf, gcdf;           // the gate stages' values of the k-step in progress
    pk2 cva, cvb, cra, crb;                        // the split's values of the piece in progress
    uint32_t ch0 = 0, ch1 = 0, cm0 = 0, cm1 = 0, cl0 = 0, cl1 = 0;
    auto up = [](uint32_t w) { float a, b; Vec8<bf16_t>::unpack2(w, a, b); return pk2{a, b}; };
    auto chunk = [&](auto g_c) {                   // CPP chunks per piece, each about what fits one MFMA's shadow
      constexpr int g = decltype(g_c)::value, jp = g / CPP, c = g % CPP;
      if constexpr (g == 0) load_gate(cur_g);
      if constexpr (c == 0) {
        cva = pk2{raw[jp].x, raw[jp].y};
        ch0 = Vec8<bf16_t>::pack(cva.x, cva.y);
      } else if constexpr (c == 1) {
        cvb = pk2{raw[jp].z, raw[jp].w};
        ch1 = Vec8<bf16_t>::pack(cvb.x, cvb.y);
      } else if constexpr (c == 2) {
        cra = cva - up(ch0);                                // (exact: the remainder of a round-to-nearest)
        cm0 = Vec8<bf16_t>::pack(cra.x, cra.y);
      } else if constexpr (c == 3) {
        crb = cvb - up(ch1);
        cm1 = Vec8<bf16_t>::pack(crb.x, crb.y);
      } else if constexpr (c == 4) {
        const pk2 la = cra - up(cm0);
        cl0 = Vec8<bf16_t>::pack(la.x, la.y);
      } else if constexpr (c == 5) {
        const pk2 lb = crb - up(cm1);
        cl1 = Vec8<bf16_t>::pack(lb.x, lb.y);
      } else if constexpr (c == 6) {
        unsigned char* d = cbase + nxt + (jp / CPL) * 8 * XS + (jp % CPL) * 256;
#ifdef EXP_FS_NOSPLIT
        *reinterpret_cast<u32x2s*>(d) = u32x2s{__float_as_uint(cva.x), __float_as_uint(cva.y)};
        *reinterpret_cast<u32x2s*>(d + PLANE) = u32x2s{__float_as_uint(cvb.x), __float_as_uint(cvb.y)};
        *reinterpret_cast<u32x2s*>(d + 2 * PLANE) = u32x2s{__float_as_uint(cva.x) + 1u, __float_as_uint(cvb.y)};
#else
        *reinterpret_cast<u32x2s*>(d) = u32x2s{ch0, ch1};
        *reinterpret_cast<u32x2s*>(d + PLANE) = u32x2s{cm0, cm1};
        *reinterpret_cast<u32x2s*>(d + 2 * PLANE) = u32x2s{cl0, cl1};
#endif
      } else {
        fetch_piece(t + 2 * G, jp);
#ifndef EXP_FS_NOSTORE
        if constexpr (g == NCHK - 1 && !SPREAD_STORES) store_pend();
#endif
      }
    };
    auto gate_stage = [&](auto s_c, auto i_c) {      // common.h gelu_erf_grad / silu_grad in six stages, values 2 s, 2 s + 1
      constexpr int s = decltype(s_c)::value, i = decltype(i_c)::value;
      if constexpr (GATE == 1) {
        if constexpr (i == 0) {
          const f32x4 gv = gq[s / 2];
          gx = (s & 1) ? pk2{gv.z, gv.w} : pk2{gv.x, gv.y};
          gsa = pk2{fminf(fabsf(gx.x), 5.656854249f), fminf(fabsf(gx.y), 5.656854249f)};
          const pk2 tt = gx * gx * -0.72134752044448170f;
          gpdf = pk2{__builtin_amdgcn_exp2f(tt.x), __builtin_amdgcn_exp2f(tt.y)} * 0.3989422804014327f;
        } else if constexpr (i == 1) {
          gq2 = pk2{-2.855192741e-06f, -2.855192741e-06f};
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{3.960562235e-05f, 3.960562235e-05f});
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{-1.871826931e-04f, -1.871826931e-04f});
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{-1.347308812e-04f, -1.347308812e-04f});
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{7.060847394e-03f, 7.060847394e-03f});
        } else if constexpr (i == 2) {
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{-5.249462857e-02f, -5.249462857e-02f});
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{-4.592086259e-01f, -4.592086259e-01f});
          gq2 = __builtin_elementwise_fma(gq2, gsa, pk2{-1.151105166e+00f, -1.151105166e+00f});
          gq2 = gq2 * gsa;
        } else if constexpr (i == 3) {
          ghalf = pk2{__builtin_amdgcn_exp2f(gq2.x), __builtin_amdgcn_exp2f(gq2.y)} * 0.5f;
        } else if constexpr (i == 4) {
          gcdf = pk2{gx.x < 0.f ? ghalf.x : 1.0f - ghalf.x, gx.y < 0.f ? ghalf.y : 1.0f - ghalf.y};
        } else {
          const pk2 g2 = __builtin_elementwise_fma(gx, gpdf, gcdf);
          pend[0][2 * s] *= g2.x; pend[0][2 * s + 1] *= g2.y;
        }
      } else if constexpr (GATE == 2) {
        if constexpr (i == 0) {
          const f32x4 gv = gq[s / 2];
          gx = (s & 1) ? pk2{gv.z, gv.w} : pk2{gv.x, gv.y};
          const pk2 tt = gx * -1.4426950408889634f;
          gsa = pk2{__builtin_amdgcn_exp2f(tt.x), __builtin_amdgcn_exp2f(tt.y)} + 1.0f;
        } else if constexpr (i == 1) {
          gpdf = pk2{__builtin_amdgcn_rcpf(gsa.x), __builtin_amdgcn_rcpf(gsa.y)};      // sigmoid
        } else if constexpr (i == 2) {
          const pk2 g2 = gpdf * (gx * (1.0f - gpdf) + 1.0f);
          pend[0][2 * s] *= g2.x; pend[0][2 * s + 1] *= g2.y;
        }
      }
    };
    static_for_wgs<NKS>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
#ifdef EXP_FS_STAMPS
      if constexpr (s == 0) { c1 = clock64(); }
      if constexpr (s == HEAD) { c2 = clock64(); }
      if constexpr (s == HEAD + NP) { c3 = clock64(); }
#endif
      if constexpr (s + 1 < NKS) {
#pragma unroll
        for (int q = 0; q < 3; ++q) xf[(s + 1) & 1][q] = *reinterpret_cast<const u32x4*>(cur + q * PLANE + 32 * (s + 1));
      }
      __builtin_amdgcn_sched_barrier(0);           // (the scheduler would sink each read to just before its MFMA)
      static_for_wgs<6>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value, slot = s * 6 + i;
        // the six products of a k-step, smallest terms first: (w lo, x hi) (w hi, x lo) (w mid, x mid) (w mid, x hi)
        // (w hi, x mid) (w hi, x hi)
        constexpr int wq = i == 0 ? 2 : (i == 2 || i == 3) ? 1 : 0;
        constexpr int xq = i == 1 ? 2 : (i == 2 || i == 4) ? 1 : 0;
#ifdef EXP_FS_ONEMFMA
        if constexpr (i == 5)
#endif
        {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) acc[ct][s % NPAR] = mfma_bf16(wf[wq][ct][s], xf[s & 1][xq], acc[ct][s % NPAR]);
        }
        if constexpr (GATE != 0 && s < HEAD && 2 * s + 1 < 16) {
          gate_stage(s_c, i_c);
          __builtin_amdgcn_sched_barrier(0);       // (pinned: left alone, the scheduler gathers the MFMAs of a k-step)
        }
        if constexpr (slot >= S0 && slot < S0 + SPAN) {
          constexpr int g0 = (slot - S0) * NCHK / SPAN, g1 = (slot + 1 - S0) * NCHK / SPAN;
          if constexpr (g1 > g0) {
            static_for_wgs<g1 - g0>([&](auto d_c) { chunk(std::integral_constant<int, g0 + decltype(d_c)::value>{}); });
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#ifndef EXP_FS_NOSTORE
        if constexpr (SPREAD_STORES && !LDS_OUT && slot >= S0 + SPAN && slot - (S0 + SPAN) < NCT * 4) {
          store_piece(slot - (S0 + SPAN));
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SPREAD_STORES && LDS_OUT && slot >= S0 + SPAN - 2 && slot - (S0 + SPAN - 2) <= NCT * 4 + 1) {
          constexpr int q = slot - (S0 + SPAN - 2);          // read piece q, store piece q - 2 (its read is two slots old)
          if constexpr (q < NCT * 4) out_read(q);
          if constexpr (q > 1) out_store(q - 2);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      });
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x16 o = acc[ct][0];
      if constexpr (NPAR == 2) o = o + acc[ct][1];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
        if constexpr (BIAS && !LDS_OUT)            // (last, as in the exact kernels: the products' small terms were summed first)
          v = v + *reinterpret_cast<const f32x4*>(lds_bias + (wave * NCT + ct) * 32 + 8 * g + 4 * h);
        if constexpr (LDS_OUT) *reinterpret_cast<f32x4*>(out_w + (ct * 32 + r) * OS + 8 * g + 4 * h) = v;
        else { pend[ct][4 * g] = v.x; pend[ct][4 * g + 1] = v.y; pend[ct][4 * g + 2] = v.z; pend[ct][4 * g + 3] = v.w; }
      }
    }
    pend_y = cur_y;
    pend_row0 = (uint32_t)t * 32u;
#ifdef EXP_FS_STAMPS
    { const long long c4 = clock64(); st_bar += c1 - c0; st_head += c2 - c1; st_commit += c3 - c2; st_tail += c4 - c3; ++st_n; }
#endif
  }
#ifdef EXP_FS_STAMPS
  if (blockIdx.x == 7 && lane == 0)
    printf("wave %d: %lld tiles, per tile: barrier %lld head %lld commit %lld tail %lld; total %lld clocks\n", wave, st_n, st_bar / st_n,
           st_head / st_n, st_commit / st_n, st_tail / st_n, (long long)(clock64() - st_t0));
#endif
#pragma unroll
  for (int e = 0; e < 16; ++e) gate_apply(e);
  store_pend();
}

} }  // namespace segger::(anonymous)

extern "C" int segger_f32_split_planes(const float* w, int32_t rows, int32_t cols, int32_t transpose, void* planes,
                                       segger_stream_t stream) {
  SEGGER_REQUIRE(rows >= 0 && cols >= 0, "segger_f32_split_planes: negative size");
  const int64_t n = (int64_t)rows * cols;
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(w && planes, "segger_f32_split_planes: NULL pointer");
  SEGGER_REQUIRE(n < (int64_t)0x7fffffff * 256, "segger_f32_split_planes: matrix too large");
  hipLaunchKernelGGL(f32_split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, rows, cols,
                     transpose, static_cast<bf16_t*>(planes));
  SEGGER_LAUNCH_CHECK("f32_split_planes_kernel");
  return SEGGER_OK;
}

extern "C" int segger_f32_split_planes_many(const segger_planes_job* jobs, int32_t n_jobs, segger_stream_t stream) {
  SEGGER_REQUIRE(jobs && n_jobs > 0 && n_jobs <= SEGGER_PLANES_MAX_JOBS, "segger_f32_split_planes_many: 1..32 jobs");
  PlanesJobs all{};
  int64_t most = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const segger_planes_job& j = jobs[i];
    SEGGER_REQUIRE(j.rows >= 0 && j.cols >= 0, "segger_f32_split_planes_many: negative size");
    const int64_t n = (int64_t)j.rows * j.cols;
    SEGGER_REQUIRE(n == 0 || (j.w && j.planes), "segger_f32_split_planes_many: NULL pointer");
    SEGGER_REQUIRE(n < (int64_t)0x7fffffff * 256, "segger_f32_split_planes_many: matrix too large");
    all.job[i] = j;
    if (n > most) most = n;
  }
  if (most == 0) return SEGGER_OK;
  hipLaunchKernelGGL(f32_split_planes_many_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n_jobs), dim3(256), 0,
                     (hipStream_t)stream, all);
  SEGGER_LAUNCH_CHECK("f32_split_planes_many_kernel");
  return SEGGER_OK;
}

extern "C" int segger_linear_wgrad_f32_split_supported(int32_t m_out, int32_t k_in) { return wgrad_f32_split_shape_ok(m_out, k_in); }

extern "C" int segger_linear_fwd_f32_split_supported(int32_t k_in, int32_t m_out) {
  return (k_in == 128 && m_out > 0 && m_out % kCH == 0) || (k_in == 384 && m_out == 128);
}

static int split_fwd_launch(const float* x, int64_t ldx, const void* w3, const float* bias, const float* rowbias, int64_t ld_rb,
                            const int32_t* rowidx, float* y, int64_t ldy, int64_t n_rows, int32_t k_in, int32_t m_out,
                            segger_stream_t stream, const float* gate = nullptr, int64_t ld_gate = 0, int gate_kind = 0);

extern "C" int segger_linear_fwd_f32_split(const float* x, int64_t ldx, const void* w3, const float* bias, float* y, int64_t ldy,
                                           int64_t n_rows, int32_t k_in, int32_t m_out, segger_stream_t stream) {
  return split_fwd_launch(x, ldx, w3, bias, nullptr, 0, nullptr, y, ldy, n_rows, k_in, m_out, stream);
}

extern "C" int segger_linear_fwd_f32_split_rowbias(const float* x, int64_t ldx, const void* w3, const float* rowbias, int64_t ld_rb,
                                                   const int32_t* rowidx, float* y, int64_t ldy, int64_t n_rows, int32_t k_in,
                                                   int32_t m_out, segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows == 0 || (rowbias && rowidx && aligned16(rowbias) && ld_rb >= m_out && ld_rb % 4 == 0),
                 "segger_linear_fwd_f32_split_rowbias: table NULL, misaligned or ld < m_out");
  return split_fwd_launch(x, ldx, w3, nullptr, rowbias, ld_rb, rowidx, y, ldy, n_rows, k_in, m_out, stream);
}

extern "C" int segger_linear_fwd_f32_act(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                                         float* y_act, int64_t ld_yact, int32_t act_kind, int64_t n_rows, int32_t k_in,
                                         int32_t m_out, segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows >= 0 && (act_kind == 1 || act_kind == 2), "segger_linear_fwd_f32_act: act_kind 1 (GELU) or 2 (SiLU)");
  if (n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(x && w && y && y_act && aligned16(x) && aligned16(w) && aligned16(y) && aligned16(y_act) && (!bias || aligned16(bias)) &&
                     ldx >= k_in && ldy >= m_out && ld_yact >= m_out && ldx % 4 == 0 && ldy % 4 == 0 && ld_yact % 4 == 0,
                 "segger_linear_fwd_f32_act: NULL pointer or rows not 16-byte aligned");
  if (!((k_in == 64 || k_in == 128 || k_in == 256) && m_out > 0 && m_out % 64 == 0)) {
    set_error("segger_linear_fwd_f32_act: k_in=%d m_out=%d not supported (k_in 64 / 128 / 256, m_out %% 64 == 0)", k_in, m_out);
    return SEGGER_EUNSUPPORTED;
  }
  return linear_f32_launch(x, ldx, w, bias, y, ldy, n_rows, k_in, m_out, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, 0, 0,
                           y_act, ld_yact, act_kind);
}

extern "C" int segger_linear_fwd_f32_gate(const float* x, int64_t ldx, const void* w, int32_t w_is_planes, const float* gate,
                                          int64_t ld_gate, int32_t gate_kind, float* y, int64_t ldy, int64_t n_rows, int32_t k_in,
                                          int32_t m_out, segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows >= 0 && (gate_kind == 1 || gate_kind == 2), "segger_linear_fwd_f32_gate: gate_kind 1 (GELU) or 2 (SiLU)");
  if (n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(x && w && y && gate && aligned16(x) && aligned16(w) && aligned16(y) && aligned16(gate) && ldx >= k_in &&
                     ldy >= m_out && ld_gate >= m_out && ldx % 4 == 0 && ldy % 4 == 0 && ld_gate % 4 == 0,
                 "segger_linear_fwd_f32_gate: NULL pointer or rows not 16-byte aligned");
  if (w_is_planes) return split_fwd_launch(x, ldx, w, nullptr, nullptr, 0, nullptr, y, ldy, n_rows, k_in, m_out, stream, gate, ld_gate, gate_kind);
  if (!((k_in == 64 || k_in == 128 || k_in == 256) && m_out > 0 && m_out % 64 == 0)) {
    set_error("segger_linear_fwd_f32_gate: k_in=%d m_out=%d not supported (k_in 64 / 128 / 256, m_out %% 64 == 0)", k_in, m_out);
    return SEGGER_EUNSUPPORTED;
  }
  return linear_f32_launch(x, ldx, w, nullptr, y, ldy, n_rows, k_in, m_out, (hipStream_t)stream, nullptr, nullptr, 0, gate, ld_gate,
                           gate_kind);
}

static int split_fwd_launch(const float* x, int64_t ldx, const void* w3, const float* bias, const float* rowbias, int64_t ld_rb,
                            const int32_t* rowidx, float* y, int64_t ldy, int64_t n_rows, int32_t k_in, int32_t m_out,
                            segger_stream_t stream, const float* gate, int64_t ld_gate, int gate_kind) {
  SEGGER_REQUIRE(n_rows >= 0, "segger_linear_fwd_f32_split: negative size");
  if (!segger_linear_fwd_f32_split_supported(k_in, m_out)) {
    set_error("segger_linear_fwd_f32_split: k_in=%d m_out=%d not supported (128 -> multiple of 64, 384 -> 128)", k_in, m_out);
    return SEGGER_EUNSUPPORTED;
  }
  if (n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(x && w3 && y, "segger_linear_fwd_f32_split: NULL pointer");
  SEGGER_REQUIRE(aligned16(x) && aligned16(w3) && aligned16(y) && (!bias || aligned16(bias)) && ldx >= k_in && ldy >= m_out &&
                     ldx % 4 == 0 && ldy % 4 == 0, "segger_linear_fwd_f32_split: rows (and the bias) must be 16-byte aligned");
  const int64_t nb = (n_rows + 127) / 128;
  SEGGER_REQUIRE(nb <= 0x7fffffffLL, "segger_linear_fwd_f32_split: too many rows");
  SplitParams p{x, ldx, static_cast<const bf16_t*>(w3), bias, y, ldy, n_rows, m_out, rowbias, rowidx, ld_rb, gate, ld_gate, gate_kind};
#ifndef SEGGER_FS_WRES
#define SEGGER_FS_WRES 1                   // the two K * M = 49152 shapes on the W-resident kernel
#endif
  const int64_t n_tiles = (n_rows + 31) / 32;
  const unsigned wres_grid = (unsigned)std::min<int64_t>(n_tiles, device_cu_count());
  // (the W-resident kernels address x, y and the gate with 31-bit byte offsets)
  // (SEGGER_AMD_F32_WRES=0 in the environment: the chunked kernels for every shape -- A/B runs without a rebuild)
  static const bool wres_env = [] { const char* e = getenv("SEGGER_AMD_F32_WRES"); return !(e && e[0] == '0'); }();
  const bool wres_ok = SEGGER_FS_WRES && wres_env && !rowbias && n_rows * ldx * 4 < 0x7ffff000LL && n_rows * ldy * 4 < 0x7ffff000LL &&
                       (!gate || n_rows * ld_gate * 4 < 0x7ffff000LL);
#define WRES(KK, MM, GG, BB) hipLaunchKernelGGL((linear_f32_split_wres_kernel<KK, MM, GG, BB>), dim3(wres_grid), dim3(256), 0, \
                                                (hipStream_t)stream, p, (int)n_tiles)
  if (wres_ok && k_in == 384 && m_out == 128 && !bias && gate && gate_kind == 1) WRES(384, 128, 1, false);
  else if (wres_ok && k_in == 384 && m_out == 128 && !bias && gate) WRES(384, 128, 2, false);
  else if (wres_ok && k_in == 384 && m_out == 128 && !bias) WRES(384, 128, 0, false);
  else if (wres_ok && k_in == 384 && m_out == 128 && !gate) WRES(384, 128, 0, true);
  else if (wres_ok && k_in == 128 && m_out == 384 && !gate && bias) WRES(128, 384, 0, true);
  else if (wres_ok && k_in == 128 && m_out == 384 && !gate) WRES(128, 384, 0, false);
#undef WRES
  else if (k_in == 128) hipLaunchKernelGGL((linear_f32_split_kernel<128>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((linear_f32_split_kernel<384>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  SEGGER_LAUNCH_CHECK("linear_f32_split_kernel");
  return SEGGER_OK;
}
