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
#include "common.h"

namespace segger {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct SplitParams {
  const float* x; int64_t ldx;
  const bf16_t* w3;          // [3][m_out][K]
  const float* bias;
  float* y; int64_t ldy;
  int64_t n_rows;
  int m_out;
};

// 8 consecutive floats -> their hi / mid / lo bf16 parts (round to nearest each time: the remainders are exact in fp32)
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
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
template <int K>
__global__ __launch_bounds__(256, 2) void linear_f32_split_kernel(SplitParams p) {
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
      acc0 = mfma_bf16(wl0, xh[s], acc0); acc1 = mfma_bf16(wl1, xh[s], acc1);        // smallest terms first
      acc0 = mfma_bf16(wh0, xl[s], acc0); acc1 = mfma_bf16(wh1, xl[s], acc1);
      acc0 = mfma_bf16(wm0, xm[s], acc0); acc1 = mfma_bf16(wm1, xm[s], acc1);
      acc0 = mfma_bf16(wm0, xh[s], acc0); acc1 = mfma_bf16(wm1, xh[s], acc1);
      acc0 = mfma_bf16(wh0, xm[s], acc0); acc1 = mfma_bf16(wh1, xm[s], acc1);
      acc0 = mfma_bf16(wh0, xh[s], acc0); acc1 = mfma_bf16(wh1, xh[s], acc1);
    }
  };
  auto store_tile = [&](const f32x16& acc, int c0, int ct) {
    if (!row_ok) return;
    float* yr = p.y + row * p.ldy + c0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = ct * 32 + 8 * g + 4 * h;
      f32x4 v = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      if (p.bias) v = v + *reinterpret_cast<const f32x4*>(p.bias + c0 + col);
      *reinterpret_cast<f32x4*>(yr + col) = v;
    }
  };

  f32x16 acc[NACC];
  if constexpr (NS == 1) {
    load_x(0);
    const int n_chunks = M / kCH;
    for (int c = 0; c < n_chunks; ++c) {
      __syncthreads();                                       // every wave has left the previous chunk's reads
      stage_w(c * kCH, 0);
      __syncthreads();
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
#pragma unroll 1
    for (int slice = 0; slice < NS; ++slice) {
      load_x(slice);
#pragma unroll
      for (int c = 0; c < NACC / 2; ++c) {
        __syncthreads();
        stage_w(c * kCH, slice);
        __syncthreads();
        chunk_mma(acc[2 * c], acc[2 * c + 1]);
      }
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) store_tile(acc[a], (a / 2) * kCH, a & 1);
  }
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_linear_fwd_f32_split_supported(int32_t k_in, int32_t m_out) {
  return (k_in == 128 && m_out > 0 && m_out % kCH == 0) || (k_in == 384 && m_out == 128);
}

extern "C" int segger_linear_fwd_f32_split(const float* x, int64_t ldx, const void* w3, const float* bias, float* y, int64_t ldy,
                                           int64_t n_rows, int32_t k_in, int32_t m_out, segger_stream_t stream) {
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
  SplitParams p{x, ldx, static_cast<const bf16_t*>(w3), bias, y, ldy, n_rows, m_out};
  if (k_in == 128) hipLaunchKernelGGL((linear_f32_split_kernel<128>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((linear_f32_split_kernel<384>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  SEGGER_LAUNCH_CHECK("linear_f32_split_kernel");
  return SEGGER_OK;
}
