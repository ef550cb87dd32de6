// The gene-embedding half of the first layer's projections as a per-gene table, and its backward.
//
// ISTEncoder's layer-0 input of a transcript is gelu(cat(E[gene], pe)) (reference ist_encoder.py:312-320) and the first
// SkipGAT projects it with three stacked nn.Linear maps (GATv2Conv.lin_l / lin_r of tx-neighbors-tx, lin_l of
// tx-belongs-bd; ist_encoder.py:111-124).  The embedding half of that product depends on a row only through its gene:
//     T[g, :] = gelu(E[g, :]) Wa^T + b            Wa = W[:, 0:D]   ([n_genes, M]: a few hundred rows)
// and segger_linear_fwd_rowbias adds T[gene[row]] in the epilogue of the GEMM over the positional half Wc = W[:, D:2D].
// Round 4 formed T (and the gradients that flow back through it) with torch ops: gelu, cat of the weights, a vendor GEMM,
// cat of the biases, add, casts -- and in the backward two more vendor GEMMs (51 us for a 384 x 256 x 128 product), a
// reduction, slice gradients (fills + copies) and adds: ~25 launches, 0.25 ms of the 13 ms C2 step.  Here:
//   segger_gene_table_fwd   one launch: T in the compute dtype, Wc and Wc^T cast to the compute dtype
//   segger_gene_table_bwd   one launch: dE = gelu'(E) * (dT Wa),  dW_i = [dT^T gelu(E) | dWc] (both halves of every weight's
//                           gradient, written in place),  db_i = column sums of dT
// dT [n_genes, M] fp32 is the by-gene row sum of dY (segger_segment_rowsum), dWc [M, D] fp32 comes from the MFMA weight-
// gradient kernel.  These are small dense products (G x M x D = 256 x 384 x 128): plain fp32 FMA tiles through LDS, fp32
// throughout (the masters are fp32; torch computed them in fp32 as well).
#include "common.h"

namespace segger {
namespace {

// a 256-thread workgroup owns a 32 x 64 tile of the product; 64 k per staged step: the products are latency-bound (16-24
// workgroups, each a chain of load -> barrier -> FMA steps), so few long steps with 24 loads in flight per thread beat many
// short ones (16 k per step: 79 us for the backward at G = 256, M = 384, D = 128)
constexpr int kTI = 32, kTJ = 64, kTK = 64;

struct GeneTable {
  const float* table; int G, D, n_w, M;
  const float* w[4]; int64_t ld_w[4]; int m_off[5];
  const float* b[4];
  void* tab; int64_t ld_tab; void* wc; void* wc_t;
  const float* g_tab; const float* g_wc;
  float* g_table; float* g_w[4]; float* g_b[4];
  int nb_a, nb_b, nb_c;                              // block ranges of a launch
};

// stacked row m of the weights -> (pointer to its row, which matrix)
__device__ __forceinline__ const float* w_row(const GeneTable& p, int m, int& which) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) if (k < p.n_w && m >= p.m_off[k]) i = k;
  which = i;
  return p.w[i] + (int64_t)(m - p.m_off[i]) * p.ld_w[i];
}

// C[i, j] = sum_k A(i, k) B(k, j) for the tile (i0, j0); A / B are element functors (0 outside the matrix is their job)
template <typename FA, typename FB, typename FE>
__device__ __forceinline__ void tile_gemm(int K, int i0, int j0, FA a, FB b, FE epilogue) {
  __shared__ float As[kTI][kTK + 1];
  __shared__ float Bs[kTK][kTJ + 1];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;          // rows 2 ti, 2 ti + 1; columns tj + 16 c
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += kTK) {
    float av[kTI * kTK / 256], bv[kTK * kTJ / 256];                    // all of the step's loads first, then the LDS writes
#pragma unroll
    for (int q = 0; q < kTI * kTK / 256; ++q) {
      const int e = tid + 256 * q;
      av[q] = a(i0 + e / kTK, k0 + e % kTK);
    }
#pragma unroll
    for (int q = 0; q < kTK * kTJ / 256; ++q) {
      const int e = tid + 256 * q;
      bv[q] = b(k0 + e / kTJ, j0 + e % kTJ);
    }
#pragma unroll
    for (int q = 0; q < kTI * kTK / 256; ++q) {
      const int e = tid + 256 * q;
      As[e / kTK][e % kTK] = av[q];
    }
#pragma unroll
    for (int q = 0; q < kTK * kTJ / 256; ++q) {
      const int e = tid + 256 * q;
      Bs[e / kTJ][e % kTJ] = bv[q];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kTK; ++kk) {
      const float a0 = As[2 * ti][kk], a1 = As[2 * ti + 1][kk];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float bv = Bs[kk][tj + 16 * c];
        acc[0][c] = fmaf(a0, bv, acc[0][c]);
        acc[1][c] = fmaf(a1, bv, acc[1][c]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) epilogue(i0 + 2 * ti + r, j0 + tj + 16 * c, acc[r][c]);
}

template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { p->v = (uint16_t)(Vec8<bf16_t>::pack(v, 0.f) & 0xffffu); }
template <> __device__ __forceinline__ void st1<f16_t>(f16_t* p, float v) { p->v = (uint16_t)(Vec8<f16_t>::pack(v, 0.f) & 0xffffu); }

// ---- forward: [ T tiles | cast of the positional half (Wc, Wc^T) ] ----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gene_table_fwd_kernel(GeneTable p) {
  const int blk = blockIdx.x;
  const int D = p.D, M = p.M, G = p.G;
  if (blk < p.nb_a) {
    const int tiles_j = (M + kTJ - 1) / kTJ;
    const int i0 = (blk / tiles_j) * kTI, j0 = (blk % tiles_j) * kTJ;
    tile_gemm(D, i0, j0,
              [&](int g, int d) { return (g < G && d < D) ? gelu_erf(p.table[(int64_t)g * D + d]) : 0.f; },
              [&](int d, int m) { int wi; return (m < M && d < D) ? w_row(p, m, wi)[d] : 0.f; },
              [&](int g, int m, float v) {
                if (g >= G || m >= M) return;
                int wi;
                w_row(p, m, wi);
                if (p.b[wi]) v += p.b[wi][m - p.m_off[wi]];
                st1(static_cast<T*>(p.tab) + (int64_t)g * p.ld_tab + m, v);
              });
  } else {
    // Wc[m, d] = W[m, D + d] in the compute dtype, and its transpose [D, M]
    const int64_t e = (int64_t)(blk - p.nb_a) * 256 + threadIdx.x;
    if (e >= (int64_t)M * D) return;
    const int m = (int)(e / D), d = (int)(e % D);
    int wi;
    const float v = w_row(p, m, wi)[D + d];
    if (p.wc) st1(static_cast<T*>(p.wc) + (int64_t)m * D + d, v);
    if (p.wc_t) st1(static_cast<T*>(p.wc_t) + (int64_t)d * M + m, v);
  }
}

// ---- backward: [ dE tiles | dWa tiles | copy of dWc + bias sums ] ------------------------------------------------------
__global__ __launch_bounds__(256) void gene_table_bwd_kernel(GeneTable p) {
  const int blk = blockIdx.x;
  const int D = p.D, M = p.M, G = p.G;
  const int tiles_d = (D + kTJ - 1) / kTJ;
  if (blk < p.nb_a) {
    // dE[g, d] = gelu'(E[g, d]) * sum_m dT[g, m] Wa[m, d]
    const int i0 = (blk / tiles_d) * kTI, j0 = (blk % tiles_d) * kTJ;
    tile_gemm(M, i0, j0,
              [&](int g, int m) { return (g < G && m < M) ? p.g_tab[(int64_t)g * M + m] : 0.f; },
              [&](int m, int d) { int wi; return (m < M && d < D) ? w_row(p, m, wi)[d] : 0.f; },
              [&](int g, int d, float v) {
                if (g < G && d < D) p.g_table[(int64_t)g * D + d] = v * gelu_erf_grad(p.table[(int64_t)g * D + d]);
              });
  } else if (blk < p.nb_a + p.nb_b) {
    // dWa[m, d] = sum_g dT[g, m] gelu(E[g, d])  -> left half of the owning weight's gradient
    const int b2 = blk - p.nb_a;
    const int i0 = (b2 / tiles_d) * kTI, j0 = (b2 % tiles_d) * kTJ;
    tile_gemm(G, i0, j0,
              [&](int m, int g) { return (m < M && g < G) ? p.g_tab[(int64_t)g * M + m] : 0.f; },
              [&](int g, int d) { return (g < G && d < D) ? gelu_erf(p.table[(int64_t)g * D + d]) : 0.f; },
              [&](int m, int d, float v) {
                if (m >= M || d >= D) return;
                int wi;
                w_row(p, m, wi);
                if (p.g_w[wi]) p.g_w[wi][(int64_t)(m - p.m_off[wi]) * (2 * D) + d] = v;
              });
  } else {
    // right half of every weight's gradient = dWc (from the MFMA weight-gradient kernel); bias gradient = column sums of dT
    const int64_t e = (int64_t)(blk - p.nb_a - p.nb_b) * 256 + threadIdx.x;
    if (e < (int64_t)M * D) {
      const int m = (int)(e / D), d = (int)(e % D);
      int wi;
      w_row(p, m, wi);
      if (p.g_w[wi] && p.g_wc) p.g_w[wi][(int64_t)(m - p.m_off[wi]) * (2 * D) + D + d] = p.g_wc[e];
    } else if (e < (int64_t)M * D + M) {
      const int m = (int)(e - (int64_t)M * D);
      int wi;
      w_row(p, m, wi);
      if (p.g_b[wi]) {
        float s0 = 0.f, s1 = 0.f;
        int g = 0;
        for (; g + 1 < G; g += 2) { s0 += p.g_tab[(int64_t)g * M + m]; s1 += p.g_tab[(int64_t)(g + 1) * M + m]; }
        if (g < G) s0 += p.g_tab[(int64_t)g * M + m];
        p.g_b[wi][m - p.m_off[wi]] = s0 + s1;
      }
    }
  }
}

int fill(const segger_gene_table_args* a, bool bwd, GeneTable* out) {
  SEGGER_REQUIRE(a != nullptr, "segger_gene_table: args is NULL");
  SEGGER_REQUIRE(a->n_genes > 0 && a->D > 0 && a->n_w >= 1 && a->n_w <= 4 && a->table, "segger_gene_table: table / sizes (1..4 weights)");
  SEGGER_REQUIRE(a->dtype == SEGGER_F32 || a->dtype == SEGGER_BF16 || a->dtype == SEGGER_F16, "segger_gene_table: unknown dtype %d", a->dtype);
  GeneTable p{};
  p.table = a->table; p.G = a->n_genes; p.D = a->D; p.n_w = a->n_w;
  int m = 0;
  for (int i = 0; i < a->n_w; ++i) {
    SEGGER_REQUIRE(a->w[i] && a->m[i] > 0 && a->ld_w[i] >= 2 * (int64_t)a->D, "segger_gene_table: weight %d: NULL, empty or ld < 2 D", i);
    p.w[i] = a->w[i]; p.ld_w[i] = a->ld_w[i]; p.b[i] = a->b[i]; p.m_off[i] = m;
    p.g_w[i] = a->g_w[i]; p.g_b[i] = a->g_b[i];
    m += a->m[i];
    SEGGER_REQUIRE(m < (1 << 24), "segger_gene_table: too many output rows");
  }
  for (int i = a->n_w; i < 5; ++i) p.m_off[i] = m;
  p.M = m;
  p.tab = a->tab; p.ld_tab = a->ld_tab; p.wc = a->wc; p.wc_t = a->wc_t;
  p.g_tab = a->g_tab; p.g_wc = a->g_wc; p.g_table = a->g_table;
  if (!bwd) {
    SEGGER_REQUIRE(a->tab && a->ld_tab >= m, "segger_gene_table_fwd: tab is NULL or ld < M");
  } else {
    SEGGER_REQUIRE(a->g_tab, "segger_gene_table_bwd: g_tab is NULL");
  }
  *out = p;
  return SEGGER_OK;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_gene_table_fwd(const segger_gene_table_args* a, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GeneTable p;
  const int rc = fill(a, false, &p);
  if (rc != SEGGER_OK) return rc;
  p.nb_a = ((p.G + kTI - 1) / kTI) * ((p.M + kTJ - 1) / kTJ);
  p.nb_b = (p.wc || p.wc_t) ? (int)(((int64_t)p.M * p.D + 255) / 256) : 0;
  const dim3 grid((unsigned)(p.nb_a + p.nb_b));
  switch (a->dtype) {
    case SEGGER_F32:  hipLaunchKernelGGL((gene_table_fwd_kernel<float>), grid, dim3(256), 0, stream, p); break;
    case SEGGER_BF16: hipLaunchKernelGGL((gene_table_fwd_kernel<bf16_t>), grid, dim3(256), 0, stream, p); break;
    default:          hipLaunchKernelGGL((gene_table_fwd_kernel<f16_t>), grid, dim3(256), 0, stream, p); break;
  }
  SEGGER_LAUNCH_CHECK("gene_table_fwd_kernel");
  return SEGGER_OK;
}

extern "C" int segger_gene_table_bwd(const segger_gene_table_args* a, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GeneTable p;
  const int rc = fill(a, true, &p);
  if (rc != SEGGER_OK) return rc;
  const int tiles_d = (p.D + kTJ - 1) / kTJ;
  p.nb_a = p.g_table ? ((p.G + kTI - 1) / kTI) * tiles_d : 0;
  p.nb_b = ((p.M + kTI - 1) / kTI) * tiles_d;
  p.nb_c = (int)(((int64_t)p.M * p.D + p.M + 255) / 256);
  hipLaunchKernelGGL(gene_table_bwd_kernel, dim3((unsigned)(p.nb_a + p.nb_b + p.nb_c)), dim3(256), 0, stream, p);
  SEGGER_LAUNCH_CHECK("gene_table_bwd_kernel");
  return SEGGER_OK;
}
