// The first Linear of the positional embedder WITHOUT its 256-wide feature matrix and without its GEMM
// (reference src/segger/models/ist_encoder.py:22-31 sinusoidal_embedding, :57-79 Positional2dEmbedder.forward):
//
//   z1[row, m] = b0[m] + sum_j W0[m, j] cos(w_j p) + W0[m, half + j] sin(w_j p),      w_j = max_period^(-j / half) <= 1,
//
// with p the per-graph min-max NORMALISED coordinate of the row, so 0 <= p <= 1 and |w_j p| <= 1 for every feature: cos and
// sin are entire, their Taylor series at 0 truncated after p^12 are exact to 1/13! = 1.6e-10 on that range, and the sum over
// the 256 features can be taken FIRST:
//
//   z1[row, m] = sum_{d=0..12} c[m, d] p^d,      c[m, d] = [d = 0] b0[m] + (-1)^floor(d/2) / d! * sum_j W0[m, (d odd) half + j] w_j^d
//
// -- 13 coefficients per output channel (a [dim, 256] x [256, 13] product once per step, in float64), then 12 fused
// multiply-adds per channel and row in exact fp32 instead of 256 multiply-adds on a matrix pipe.  The weight gradient
// collapses the same way:
//
//   dW0[m, j] = sum_rows dz1[row, m] F[row, j] = sum_d V[j, d] M[m, d],   M[m, d] = sum_rows dz1[row, m] p_row^d,   db0 = M[:, 0]
//
// (V[j, d] = the Taylor coefficient of feature j at p^d): 13 moments per channel accumulated over the rows, then a
// [dim, 13] x [13, 256] product.  Used at fp32 storage, where the feature matrix [2n, 256] was 2 GB at C2, written by
// segger_posfreq and read by two exact-fp32 GEMMs (the fused 16-bit kernels generate the features as MFMA fragments instead:
// posmlp.hip).  Nothing here is approximate beyond the 1.6e-10 truncation: the arithmetic is fp32 FMA, the reductions fp32
// per lane and float64 across workgroups.
#include "common.h"

namespace segger {
namespace {

constexpr int kTerms = 13;        // p^0 .. p^12
constexpr int kCoefStride = 16;   // floats per channel in the coefficient / moment tables
constexpr int kMaxBlocks = 1024;     // partial tables the one-workgroup finish sums per entry

__device__ __forceinline__ double feature_freq(int j, int half, float log_max_period) {
  // the frequency as the reference forms it: fp32 exp of an fp32 argument (ist_encoder.py:26-28)
  return (double)expf(-log_max_period * (float)j / (float)half);
}

// 1 / d!  (a float64 division costs ~100 cycles; these two kernels are latency-sized)
__device__ __constant__ double kInvFact[kTerms] = {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320,
                                                  1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600};

// coef[m][d]: one workgroup per output channel; thread = (power d, one of 16 slices of the feature index j)
__global__ __launch_bounds__(256) void poly_coef_kernel(const float* __restrict__ w0, const float* __restrict__ b0, int fd,
                                                        float log_max_period, float* __restrict__ coef) {
  __shared__ double red[16][kCoefStride];
  const int m = blockIdx.x, d = threadIdx.x % kCoefStride, part = threadIdx.x / kCoefStride, half = fd / 2;
  double s = 0.0;
  if (d < kTerms) {
    const float* row = w0 + (int64_t)m * fd + ((d & 1) ? half : 0);      // cos block for even powers, sin block for odd ones
    for (int j = part; j < half; j += 16) {
      const double om = feature_freq(j, half, log_max_period);
      double pw = 1.0;
      for (int i = 0; i < d; ++i) pw *= om;
      s += (double)row[j] * pw;
    }
  }
  red[part][d] = s;
  __syncthreads();
  if (part == 0) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += red[q][d];
    float out = 0.f;
    if (d < kTerms) {
      t = t * (((d >> 1) & 1) ? -1.0 : 1.0) * kInvFact[d];
      if (d == 0 && b0) t += (double)b0[m];
      out = (float)t;
    }
    coef[m * kCoefStride + d] = out;
  }
}

struct PolyFwd {
  const float* pos; const int64_t* batch; const float* mins; const float* maxs;
  int64_t n; float eps;
  const float* coef;
  float* z1; float* h1; float* pn;
};

// lane = (row of the wave's row group, 4 consecutive channels): a row's DIM channels are one contiguous store of DIM * 4
// bytes by DIM / 4 adjacent lanes; a lane keeps its 4 x 13 coefficients in registers for all its rows
template <int DIM>
__global__ __launch_bounds__(256) void poly_fwd_kernel(PolyFwd p) {
  constexpr int LPR = DIM / 4, RPW = 64 / LPR;
  static_assert(DIM % 4 == 0 && 64 % LPR == 0, "DIM / 4 lanes per row must divide a wave");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rg = lane / LPR;
  float c[4][kTerms];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int d = 0; d < kTerms; ++d) c[i][d] = p.coef[(4 * sub + i) * kCoefStride + d];
  }
  const int64_t rows = 2 * p.n, step = (int64_t)gridDim.x * 4 * RPW;
  // a row's coordinate is a chain of dependent loads (batch id -> per-graph min / max, + the position): the next row's is
  // requested before this row's arithmetic and stores
  auto coord = [&](int64_t row) -> float {
    const int64_t node = row >> 1;
    const int cd = (int)(row & 1);
    const int64_t g = p.batch ? p.batch[node] : 0;
    const float lo = p.mins[2 * g + cd], hi = p.maxs[2 * g + cd];
    return (p.pos[2 * node + cd] - lo) / (hi - lo + p.eps);                    // ist_encoder.py:63-64 (eps 0) / :74
  };
  int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + rg;
  float x_next = row < rows ? coord(row) : 0.f;
  for (; row < rows; row += step) {
    const float x = x_next;
    if (row + step < rows) x_next = coord(row + step);
    float z[4], hq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float acc = c[i][kTerms - 1];
#pragma unroll
      for (int d = kTerms - 2; d >= 0; --d) acc = fmaf(acc, x, c[i][d]);
      z[i] = acc;
      hq[i] = acc * fast_rcp(1.0f + fast_exp2(-kLog2e * acc));                // SiLU
    }
    *reinterpret_cast<f32x4*>(p.z1 + row * DIM + 4 * sub) = f32x4{z[0], z[1], z[2], z[3]};
    if (p.h1) *reinterpret_cast<f32x4*>(p.h1 + row * DIM + 4 * sub) = f32x4{hq[0], hq[1], hq[2], hq[3]};
    if (p.pn && sub == 0) p.pn[row] = x;
  }
}

struct PolyMoments {
  const float* dz1; int64_t ld;
  const float* pn;
  int64_t rows;
  float* part;                // [gridDim.x][DIM][kCoefStride]
};

template <int DIM>
__global__ __launch_bounds__(256) void poly_moments_kernel(PolyMoments p) {
  constexpr int LPR = DIM / 4, RPW = 64 / LPR;
  __shared__ float red[4][DIM][kCoefStride];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rg = lane / LPR;
  float acc[4][kTerms];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int d = 0; d < kTerms; ++d) acc[i][d] = 0.f;
  }
  const int64_t step = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + rg; row < p.rows; row += step) {
    const float x = p.pn[row];
    const f32x4 dv = *reinterpret_cast<const f32x4*>(p.dz1 + row * p.ld + 4 * sub);
    const float v[4] = {dv.x, dv.y, dv.z, dv.w};
    float pw = 1.0f;
#pragma unroll
    for (int d = 0; d < kTerms; ++d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][d] = fmaf(v[i], pw, acc[i][d]);
      pw *= x;
    }
  }
  // the wave's row groups (lanes with the same `sub`), then the four waves through LDS
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int d = 0; d < kTerms; ++d) {
      float a = acc[i][d];
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
      if (rg == 0) red[wave][4 * sub + i][d] = a;
    }
  }
  __syncthreads();
  float* out = p.part + (int64_t)blockIdx.x * DIM * kCoefStride;
  for (int i = threadIdx.x; i < DIM * kCoefStride; i += 256) {
    const int m = i / kCoefStride, d = i % kCoefStride;
    out[i] = d < kTerms ? (red[0][m][d] + red[1][m][d]) + (red[2][m][d] + red[3][m][d]) : 0.f;
  }
}

// one workgroup per output channel m: M[m, :] = sum of the partial tables (float64; thread = (power d, one of 16 slices of the
// partials), slices added in a fixed order: deterministic), then dW0[m, :] = M[m, :] V^T and db0[m] = M[m, 0]
__global__ __launch_bounds__(256) void poly_wgrad_finish_kernel(const float* __restrict__ part, int nblocks, int dim, int fd,
                                                                float log_max_period, float* __restrict__ grad_w0,
                                                                float* __restrict__ grad_b0) {
  __shared__ double red[16][kCoefStride];
  __shared__ double mom[kCoefStride];
  const int m = blockIdx.x, d = threadIdx.x % kCoefStride, slice = threadIdx.x / kCoefStride;
  const int64_t width = (int64_t)dim * kCoefStride;
  const float* col = part + (int64_t)m * kCoefStride + d;
  double s0 = 0.0, s1 = 0.0;
  int b = slice;
  for (; b + 16 < nblocks; b += 32) {                      // two loads in flight per thread, 16 slices: 32 tables per round
    const float v0 = col[(int64_t)b * width], v1 = col[(int64_t)(b + 16) * width];
    s0 += (double)v0; s1 += (double)v1;
  }
  if (b < nblocks) s0 += (double)col[(int64_t)b * width];
  red[slice][d] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < kCoefStride) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    mom[threadIdx.x] = t;
  }
  __syncthreads();
  const int half = fd / 2;
  for (int f = threadIdx.x; f < fd; f += blockDim.x) {
    const int j = f % half;
    const int first = f >= half ? 1 : 0;                               // sin features: odd powers
    const double om = feature_freq(j, half, log_max_period), om2 = om * om;
    double pw = first ? om : 1.0, s = 0.0;
    for (int dd = first; dd < kTerms; dd += 2) {
      s += mom[dd] * pw * (((dd >> 1) & 1) ? -kInvFact[dd] : kInvFact[dd]);
      pw *= om2;
    }
    grad_w0[(int64_t)m * fd + f] = (float)s;
  }
  if (grad_b0 && threadIdx.x == 0) grad_b0[m] = (float)mom[0];
}

int64_t poly_grid(int64_t rows, int dim) {
  const int64_t rows_per_block = 4 * (64 / (dim / 4));
  int64_t blocks = (rows + rows_per_block * 8 - 1) / (rows_per_block * 8);       // >= 8 rows per lane group before another block pays
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  return blocks < 1 ? 1 : blocks;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_posenc_poly_supported(int32_t freq_dim, int32_t dim) {
  return (dim == 32 || dim == 64 || dim == 128) && freq_dim >= 2 && freq_dim % 2 == 0 && freq_dim <= 4096;
}

extern "C" int segger_posenc_poly_coef(const float* w0, const float* b0, int32_t freq_dim, int32_t dim, float max_period,
                                       float* coef, segger_stream_t stream) {
  SEGGER_REQUIRE(segger_posenc_poly_supported(freq_dim, dim), "segger_posenc_poly_coef: dim 32 / 64 / 128, even freq_dim <= 4096");
  SEGGER_REQUIRE(w0 && coef && max_period > 1.f, "segger_posenc_poly_coef: NULL pointer or max_period <= 1");
  hipLaunchKernelGGL(poly_coef_kernel, dim3((unsigned)dim), dim3(256), 0, (hipStream_t)stream, w0, b0, (int)freq_dim,
                     logf(max_period), coef);
  SEGGER_LAUNCH_CHECK("poly_coef_kernel");
  return SEGGER_OK;
}

extern "C" int segger_posenc_poly_fwd(const float* pos, const int64_t* batch, const float* mins, const float* maxs, int64_t n,
                                      float eps, const float* coef, int32_t dim, float* z1, float* h1, float* pn,
                                      segger_stream_t stream) {
  SEGGER_REQUIRE(n >= 0 && (dim == 32 || dim == 64 || dim == 128), "segger_posenc_poly_fwd: negative size or dim not in 32 / 64 / 128");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(pos && mins && maxs && coef && z1, "segger_posenc_poly_fwd: NULL pointer");
  SEGGER_REQUIRE(aligned16(z1) && aligned16(h1), "segger_posenc_poly_fwd: z1 / h1 must be 16-byte aligned");
  PolyFwd p{pos, batch, mins, maxs, n, eps, coef, z1, h1, pn};
  // (no partial tables here: as many workgroups as there are pairs of row batches -- a row's loads are a dependent chain
  //  batch id -> min / max -> position, which only other workgroups can hide)
  const int64_t rows_per_block = 4 * (64 / (dim / 4));
  int64_t blocks = (2 * n + 4 * rows_per_block - 1) / (4 * rows_per_block);
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks);
  if (dim == 32) hipLaunchKernelGGL((poly_fwd_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (dim == 64) hipLaunchKernelGGL((poly_fwd_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((poly_fwd_kernel<128>), grid, dim3(256), 0, (hipStream_t)stream, p);
  SEGGER_LAUNCH_CHECK("poly_fwd_kernel");
  return SEGGER_OK;
}

extern "C" size_t segger_posenc_poly_wgrad_workspace_bytes(int64_t n_rows, int32_t dim) {
  if (dim <= 0) return 16;
  return (size_t)poly_grid(n_rows > 0 ? n_rows : 1, dim) * (size_t)dim * kCoefStride * sizeof(float);
}

extern "C" int segger_posenc_poly_wgrad(const float* dz1, int64_t ld, const float* pn, int64_t n_rows, int32_t freq_dim, int32_t dim,
                                        float max_period, float* grad_w0, float* grad_b0, void* workspace, size_t workspace_bytes,
                                        segger_stream_t stream) {
  SEGGER_REQUIRE(segger_posenc_poly_supported(freq_dim, dim), "segger_posenc_poly_wgrad: dim 32 / 64 / 128, even freq_dim <= 4096");
  SEGGER_REQUIRE(n_rows >= 0 && grad_w0 && max_period > 1.f, "segger_posenc_poly_wgrad: bad size, NULL output or max_period <= 1");
  if (n_rows == 0) {
    SEGGER_HIP(hipMemsetAsync(grad_w0, 0, (size_t)dim * freq_dim * sizeof(float), (hipStream_t)stream));
    if (grad_b0) SEGGER_HIP(hipMemsetAsync(grad_b0, 0, (size_t)dim * sizeof(float), (hipStream_t)stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(dz1 && pn && aligned16(dz1) && ld >= dim && ld % 4 == 0, "segger_posenc_poly_wgrad: dz1 NULL, misaligned or ld < dim");
  const size_t need = segger_posenc_poly_wgrad_workspace_bytes(n_rows, dim);
  if (!workspace || workspace_bytes < need) {
    set_error("segger_posenc_poly_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t blocks = poly_grid(n_rows, dim);
  PolyMoments p{dz1, ld, pn, n_rows, static_cast<float*>(workspace)};
  if (dim == 32) hipLaunchKernelGGL((poly_moments_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else if (dim == 64) hipLaunchKernelGGL((poly_moments_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((poly_moments_kernel<128>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  SEGGER_LAUNCH_CHECK("poly_moments_kernel");
  hipLaunchKernelGGL(poly_wgrad_finish_kernel, dim3((unsigned)dim), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const float*>(workspace), (int)blocks, (int)dim, (int)freq_dim, logf(max_period), grad_w0, grad_b0);
  SEGGER_LAUNCH_CHECK("poly_wgrad_finish_kernel");
  return SEGGER_OK;
}
