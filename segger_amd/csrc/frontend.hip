// Encoder front end and tail as fused row-wise kernels (they were ~60 tiny
// elementwise launches per step in the first version):
//   segger_posfreq            normalise (x, y) per graph + 256-d sinusoid, written in the MLP's dtype
//   segger_embed_gelu_fwd/bwd gelu(cat(Embedding[gene], pos_emb)) and its backward, including the
//                             embedding-table gradient (LDS tables per block, no sort, no global atomics)
//   segger_l2norm_fwd/bwd     F.normalize(dim=-1)
// Reference: src/segger/models/ist_encoder.py:22-31,57-79 (sinusoid / normalisation), :312-320
// (embedding, concat, GELU), :331-332 (normalize).
#include "common.h"

namespace segger {
namespace {

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return __uint_as_float((uint32_t)p->v << 16); }
template <> __device__ __forceinline__ float ld1<f16_t>(const f16_t* p) { return static_cast<float>(__builtin_bit_cast(_Float16, p->v)); }

// ---------------------------------------------------------------------------------------------
// posfreq: out[(n*2 + c), :] = [cos(p * f_j) | sin(p * f_j)],  p = (pos[n,c] - lo) / (hi - lo + eps)
// one thread = 8 consecutive j of one (n, c): a 16-byte (32-byte for fp32) store into each half
// ---------------------------------------------------------------------------------------------
// sin / cos of the embedder's angles.  p is a min-max normalised coordinate in [0, 1] and the frequencies decay from 1, so every
// angle lies in [0, 1]: no range reduction is needed and the Taylor series to x^11 / x^12 (remainders 3e-8 / 3e-9 at |x| = 1.5,
// below fp32 rounding) replace libm's sincosf, whose ~70 instructions per angle made this kernel VALU-bound (0.43 ms for the
// 2 GB it writes at C2: 4.7 TB/s).  Outside [-1.5, 1.5] (never reached by normalised coordinates) the libm call stays.
__device__ __forceinline__ void sincos_unit(float x, float* s, float* c) {
#ifdef EXP_POSFREQ_LIBM      // A/B build: libm for every angle
  sincosf(x, s, c); return;
#endif
  if (__builtin_expect(fabsf(x) > 1.5f, 0)) { sincosf(x, s, c); return; }
  const float t = x * x;
  float ps = fmaf(t, -2.5052108385e-08f, 2.7557319224e-06f);
  ps = fmaf(ps, t, -1.9841269841e-04f);
  ps = fmaf(ps, t, 8.3333333333e-03f);
  ps = fmaf(ps, t, -1.6666666667e-01f);
  *s = fmaf(ps * t, x, x);
  float pc = fmaf(t, 2.0876756988e-09f, -2.7557319224e-07f);
  pc = fmaf(pc, t, 2.4801587302e-05f);
  pc = fmaf(pc, t, -1.3888888889e-03f);
  pc = fmaf(pc, t, 4.1666666667e-02f);
  pc = fmaf(pc, t, -0.5f);
  *c = fmaf(pc, t, 1.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void posfreq_kernel(const float* __restrict__ pos, const int64_t* __restrict__ batch,
                                                     const float* __restrict__ mins, const float* __restrict__ maxs,
                                                     int64_t n, int freq_dim, float eps, float log_max_period,
                                                     T* __restrict__ out) {
  extern __shared__ float freqs[];                       // [half]
  const int half = freq_dim / 2;
  for (int j = threadIdx.x; j < half; j += 256) freqs[j] = expf(-log_max_period * (float)j / (float)half);
  __syncthreads();
  const int per_row = half / 8;                          // threads per (n, c)
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rc = item / per_row;                     // n*2 + c
  if (rc >= 2 * n) return;
  const int j0 = (int)(item % per_row) * 8;
  const int64_t node = rc >> 1;
  const int c = (int)(rc & 1);
  const int64_t g = batch ? batch[node] : 0;
  const float lo = mins[2 * g + c], hi = maxs[2 * g + c];
  const float p = (pos[2 * node + c] - lo) / (hi - lo + eps);
  float cs[8], sn[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sincos_unit(p * freqs[j0 + k], &sn[k], &cs[k]);
  T* o = out + rc * freq_dim;
  Vec8<T>::store(o + j0, cs);
  Vec8<T>::store(o + half + j0, sn);
}

// ---------------------------------------------------------------------------------------------
// embed + concat + GELU
//   x0[n, 0:D]  = gelu(table[ids[n], :])     (table fp32 [G, D])
//   x0[n, D:2D] = gelu(pe[n, :])             (pe in T)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_gelu_fwd_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                            const T* __restrict__ pe, int64_t ld_pe, int64_t n, int D,
                                                            T* __restrict__ out, int64_t ld_out) {
  const int per_row = (2 * D) / 8;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = item / per_row;
  if (row >= n) return;
  const int c0 = (int)(item % per_row) * 8;
  float v[8];
  if (c0 < D) {
    const float* t = table + (int64_t)ids[row] * D + c0;
    Vec8<float>::load(t, v);
  } else {
    Vec8<T>::load(pe + row * ld_pe + (c0 - D), v);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
  Vec8<T>::store(out + row * ld_out + c0, v);
}

// backward, positional half + dense: g_pe[n, :] = gx0[n, D:2D] * gelu'(pe[n, :])
template <typename T>
__global__ __launch_bounds__(256) void embed_gelu_bwd_pe_kernel(const T* __restrict__ gx0, int64_t ld_g, const T* __restrict__ pe,
                                                               int64_t ld_pe, int64_t n, int D, T* __restrict__ gpe, int64_t ld_gpe) {
  const int per_row = D / 8;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = item / per_row;
  if (row >= n) return;
  const int c0 = (int)(item % per_row) * 8;
  float g[8], x[8];
  Vec8<T>::load(gx0 + row * ld_g + D + c0, g);
  Vec8<T>::load(pe + row * ld_pe + c0, x);
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] *= gelu_erf_grad(x[k]);
  Vec8<T>::store(gpe + row * ld_gpe + c0, g);
}

// ---------------------------------------------------------------------------------------------
// front join: the encoder's input of BOTH node types in one launch each way (ist_encoder.py:312-320)
//   x_tx[r] = gelu(cat(table[ids[r]], pe[r]))            r < n_tx
//   x_bd[r] = gelu(cat(xb[r], pe[n_tx + r]))             r < n_bd     (xb = lin_first['bd'] output)
// `pe` holds the positional embeddings of both types back to back (one embedder call); the backward writes its
// gradient as ONE [n_tx + n_bd, D] matrix (autograd would zero-fill and copy a full-size matrix to join two slices).
// ---------------------------------------------------------------------------------------------
struct FrontJoin {
  const float* table; const int32_t* ids; int D;
  const void* pe; int64_t ld_pe; int64_t n_tx, n_bd;
  const void* xb; int64_t ld_xb;
  void* out_tx; int64_t ld_out_tx; void* out_bd; int64_t ld_out_bd;
  const void* g_tx; int64_t ld_g_tx; const void* g_bd; int64_t ld_g_bd;
  void* g_pe; int64_t ld_g_pe; void* g_xb; int64_t ld_g_xb;
};

template <typename T>
__global__ __launch_bounds__(256) void front_join_fwd_kernel(FrontJoin p) {
  const int D = p.D, per_row = (2 * D) / 8;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = item / per_row;
  if (row >= p.n_tx + p.n_bd) return;
  const int c0 = (int)(item % per_row) * 8;
  const bool is_bd = row >= p.n_tx;
  const int64_t r = is_bd ? row - p.n_tx : row;
  float v[8];
  if (c0 >= D) Vec8<T>::load(static_cast<const T*>(p.pe) + row * p.ld_pe + (c0 - D), v);
  else if (is_bd) Vec8<T>::load(static_cast<const T*>(p.xb) + r * p.ld_xb + c0, v);
  else Vec8<float>::load(p.table + (int64_t)p.ids[r] * D + c0, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
  T* o = is_bd ? static_cast<T*>(p.out_bd) + r * p.ld_out_bd : static_cast<T*>(p.out_tx) + r * p.ld_out_tx;
  Vec8<T>::store(o + c0, v);
}

// items: [ (n_tx + n_bd) positional halves | n_bd dense halves ], D/8 threads each
template <typename T>
__global__ __launch_bounds__(256) void front_join_bwd_kernel(FrontJoin p) {
  const int D = p.D, per_row = D / 8;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t row = item / per_row;
  const int c0 = (int)(item % per_row) * 8;
  const int64_t n_all = p.n_tx + p.n_bd;
  if (row >= n_all + p.n_bd) return;
  float g[8], x[8];
  if (row < n_all) {                                   // positional half of either type
    const bool is_bd = row >= p.n_tx;
    const T* gsrc = is_bd ? static_cast<const T*>(p.g_bd) + (row - p.n_tx) * p.ld_g_bd : static_cast<const T*>(p.g_tx) + row * p.ld_g_tx;
    Vec8<T>::load(gsrc + D + c0, g);
    Vec8<T>::load(static_cast<const T*>(p.pe) + row * p.ld_pe + c0, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] *= gelu_erf_grad(x[k]);
    Vec8<T>::store(static_cast<T*>(p.g_pe) + row * p.ld_g_pe + c0, g);
  } else {                                             // dense half of a boundary row
    row -= n_all;
    Vec8<T>::load(static_cast<const T*>(p.g_bd) + row * p.ld_g_bd + c0, g);
    Vec8<T>::load(static_cast<const T*>(p.xb) + row * p.ld_xb + c0, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] *= gelu_erf_grad(x[k]);
    Vec8<T>::store(static_cast<T*>(p.g_xb) + row * p.ld_g_xb + c0, g);
  }
}

// backward, embedding table: gtable[g, c] = gelu'(table[g, c]) * sum_{n: ids[n] = g} gx0[n, c].
// LDS float atomics (ds_add_f32) retire about one lane every three clocks per CU on gfx950 -- a per-block LDS
// table took 0.66 ms for 1 M rows whatever the id distribution -- so the sum runs over the rows GROUPED BY GENE
// (gene_ptr / gene_rows: the CSR of ids, one radix sort per batch, shared by every step on that batch):
//   1. emb_plan_kernel: per gene, the number of chunks (<= kEmbMaxChunks, >= kEmbChunk rows each) and their
//      exclusive prefix -> chunk_ptr[G + 1];
//   2. emb_gather_kernel: block b finds its (gene, chunk) by binary search, every thread sums ITS 8 channels
//      over the chunk's rows in registers (full 16-B loads, 4 in flight), the row-lanes meet in LDS -> partial[b];
//   3. emb_finish_kernel: per gene, the <= kEmbMaxChunks partial rows in order, times gelu'(table).
// Deterministic, no atomics.
constexpr int kEmbChunk = 256;
constexpr int kEmbMaxChunks = 32;

__device__ __forceinline__ int emb_chunks_of(int64_t len) {
  if (len <= 0) return 0;
  const int64_t c = (len + kEmbChunk - 1) / kEmbChunk;
  return c > kEmbMaxChunks ? kEmbMaxChunks : (int)c;
}

__global__ __launch_bounds__(256) void emb_plan_kernel(const int64_t* __restrict__ gene_ptr, int G, int* __restrict__ chunk_ptr) {
  __shared__ int scan[256];
  __shared__ int carry;
  if (threadIdx.x == 0) { carry = 0; chunk_ptr[0] = 0; }
  __syncthreads();
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + threadIdx.x;
    int v = g < G ? emb_chunks_of(gene_ptr[g + 1] - gene_ptr[g]) : 0;
    scan[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int t = threadIdx.x >= d ? scan[threadIdx.x - d] : 0;
      __syncthreads();
      scan[threadIdx.x] += t;
      __syncthreads();
    }
    if (g < G) chunk_ptr[g + 1] = carry + scan[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 255) carry += scan[255];
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void emb_gather_kernel(const T* __restrict__ gx0, int64_t ld_g, const int64_t* __restrict__ gene_ptr,
                                                        const int32_t* __restrict__ gene_rows, const int* __restrict__ chunk_ptr,
                                                        int G, int D, float* __restrict__ partial) {
  extern __shared__ float red[];                        // [R][D]
  const int b = blockIdx.x;
  if (b >= chunk_ptr[G]) return;                         // uniform per block
  int lo = 0, hi = G;                                    // last gene with chunk_ptr[g] <= b
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_ptr[mid] <= b) lo = mid; else hi = mid;
  }
  const int g = lo;
  const int64_t beg = gene_ptr[g], len = gene_ptr[g + 1] - beg;
  const int nch = chunk_ptr[g + 1] - chunk_ptr[g], j = b - chunk_ptr[g];
  const int64_t per = (len + nch - 1) / nch;
  const int64_t p0 = beg + j * per, p1 = (p0 + per < beg + len) ? p0 + per : beg + len;
  const int P = D / 8, R = blockDim.x / P;
  const int q = threadIdx.x % P, rl = threadIdx.x / P;
  const T* base = gx0 + q * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t p = p0 + rl;
  for (; p + 3 * (int64_t)R < p1; p += 4 * (int64_t)R) {
    const int64_t r0 = gene_rows[p], r1 = gene_rows[p + R], r2 = gene_rows[p + 2 * R], r3 = gene_rows[p + 3 * R];
    float v0[8], v1[8], v2[8], v3[8];
    Vec8<T>::load(base + r0 * ld_g, v0);
    Vec8<T>::load(base + r1 * ld_g, v1);
    Vec8<T>::load(base + r2 * ld_g, v2);
    Vec8<T>::load(base + r3 * ld_g, v3);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (v0[k] + v1[k]) + (v2[k] + v3[k]);
  }
  for (; p < p1; p += R) {
    float v[8];
    Vec8<T>::load(base + (int64_t)gene_rows[p] * ld_g, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl * D + q * 8 + k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float sum = 0.f;
    for (int r = 0; r < R; ++r) sum += red[r * D + c];
    partial[(int64_t)b * D + c] = sum;
  }
}

__global__ __launch_bounds__(256) void emb_finish_kernel(const float* __restrict__ partial, const int* __restrict__ chunk_ptr,
                                                        const float* __restrict__ table, int G, int D, float* __restrict__ gtable) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)G * D) return;
  const int g = (int)(i / D), c = (int)(i % D);
  float s = 0.f;
  for (int b = chunk_ptr[g]; b < chunk_ptr[g + 1]; ++b) s += partial[(int64_t)b * D + c];
  gtable[i] = table ? s * gelu_erf_grad(table[i]) : s;        // table == NULL: plain segmented row sum
}

// ---------------------------------------------------------------------------------------------
// L2 row normalisation: z = y / max(||y||, eps)  ;  gy = (gz - z * <z, gz>) / max(||y||, eps)
// (exact for ||y|| >= eps; below eps torch's clamp makes the op linear: gy = gz / eps)
// LPC = C/8 lanes per row.
// ---------------------------------------------------------------------------------------------
template <typename T, int LPC, bool BWD>
__global__ __launch_bounds__(256) void l2norm_kernel(const T* __restrict__ y, int64_t ld_y, const T* __restrict__ gz, int64_t ld_gz,
                                                    int64_t n, float eps, T* __restrict__ out, int64_t ld_out,
                                                    const T* __restrict__ gz2 = nullptr, int64_t ld_gz2 = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPW = 64 / LPC;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + lane / LPC;
  if (row >= n) return;
  const int c0 = (lane % LPC) * 8;
  float v[8];
  Vec8<T>::load(y + row * ld_y + c0, v);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) ss = fmaf(v[k], v[k], ss);
  const float nrm = sqrtf(lane_block_sum<LPC>(ss));
  const float inv = 1.0f / fmaxf(nrm, eps);
  if (!BWD) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= inv;
    Vec8<T>::store(out + row * ld_out + c0, v);
  } else {
    float g[8];
    Vec8<T>::load(gz + row * ld_gz + c0, g);
    if (gz2) {                                           // the incoming gradient arrives as the sum of two matrices
      float g2[8];
      Vec8<T>::load(gz2 + row * ld_gz2 + c0, g2);
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] += g2[k];
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot = fmaf(v[k] * inv, g[k], dot);
    dot = lane_block_sum<LPC>(dot);
    if (nrm < eps) dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = (g[k] - v[k] * inv * dot) * inv;
    Vec8<T>::store(out + row * ld_out + c0, g);
  }
}

template <typename T, bool BWD>
int launch_l2norm(const void* y, int64_t ld_y, const void* gz, int64_t ld_gz, int64_t n, int C, float eps, void* out,
                  int64_t ld_out, hipStream_t stream, const void* gz2 = nullptr, int64_t ld_gz2 = 0) {
#define GO(LPC)                                                                                                      \
  {                                                                                                                  \
    const int64_t rpb = 4 * (64 / LPC);                                                                              \
    hipLaunchKernelGGL((l2norm_kernel<T, LPC, BWD>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, stream,    \
                       (const T*)y, ld_y, (const T*)gz, ld_gz, n, eps, (T*)out, ld_out, (const T*)gz2, ld_gz2);      \
  }
  switch (C) {
    case 8: GO(1) break;
    case 16: GO(2) break;
    case 32: GO(4) break;
    case 64: GO(8) break;
    case 128: GO(16) break;
    default: set_error("segger_l2norm: channels=%d not supported (8,16,32,64,128)", C); return SEGGER_EUNSUPPORTED;
  }
#undef GO
  SEGGER_LAUNCH_CHECK("l2norm_kernel");
  return SEGGER_OK;
}

// Several row normalisations in one launch (segger_l2norm_many): per segment forward (gz == NULL) or backward, the
// incoming gradient in the embeddings' dtype or in fp32 (the loss head accumulates the boundary side in fp32).
constexpr int kL2MaxSegs = 4;
struct L2Batch { segger_l2norm_seg seg[kL2MaxSegs]; int32_t first_block[kL2MaxSegs + 1]; int32_t n; float eps; };

template <typename T, int LPC>
__global__ __launch_bounds__(256) void l2norm_many_kernel(L2Batch b) {
  int s = 0;
  while (s + 1 < b.n && (int)blockIdx.x >= b.first_block[s + 1]) ++s;
  const segger_l2norm_seg g = b.seg[s];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPW = 64 / LPC;
  const int64_t row = ((int64_t)(blockIdx.x - b.first_block[s]) * 4 + wave) * RPW + lane / LPC;
  if (row >= g.n) return;
  const int c0 = (lane % LPC) * 8;
  float v[8];
  Vec8<T>::load(static_cast<const T*>(g.y) + row * g.ld_y + c0, v);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) ss = fmaf(v[k], v[k], ss);
  const float nrm = sqrtf(lane_block_sum<LPC>(ss));
  const float inv = 1.0f / fmaxf(nrm, b.eps);
  T* out = static_cast<T*>(g.out) + row * g.ld_out + c0;
  if (g.gz == nullptr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= inv;
    Vec8<T>::store(out, v);
    return;
  }
  float gr[8];
  if (g.gz_f32) Vec8<float>::load(static_cast<const float*>(g.gz) + row * g.ld_gz + c0, gr);
  else Vec8<T>::load(static_cast<const T*>(g.gz) + row * g.ld_gz + c0, gr);
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) dot = fmaf(v[k] * inv, gr[k], dot);
  dot = lane_block_sum<LPC>(dot);
  if (nrm < b.eps) dot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) gr[k] = (gr[k] - v[k] * inv * dot) * inv;
  Vec8<T>::store(out, gr);
}

size_t esize(int dtype) { return dtype == SEGGER_F32 ? 4 : 2; }


// ---------------------------------------------------------------------------------------------
// Column sums of a tall matrix: out[c] = sum_n x[n, c]  (bias gradients of the projections).
// P = cols/8 lanes cover one row with 16-byte loads; a block of P * R threads (R = 256 / P rows per
// sweep) walks a contiguous row range, every thread accumulating ITS 8 columns in fp32 registers
// (4 independent loads in flight), then the R row-lanes of a column group are summed through LDS and
// the block writes one partial row; a second kernel sums the partial rows.  Deterministic, no atomics.
// ---------------------------------------------------------------------------------------------
constexpr int kColsumMaxBlocks = 1024;

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t ld, int64_t n, int cols,
                                                            int64_t rows_per_block, float* __restrict__ partial) {
  extern __shared__ float red[];                        // [R][cols]
  const int P = cols / 8, R = blockDim.x / P;
  const int q = threadIdx.x % P, rl = threadIdx.x / P;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const T* base = x + q * 8;
  int64_t row = r0 + rl;
  for (; row + 3 * (int64_t)R < r1; row += 4 * (int64_t)R) {
    float v0[8], v1[8], v2[8], v3[8];
    Vec8<T>::load(base + row * ld, v0);
    Vec8<T>::load(base + (row + R) * ld, v1);
    Vec8<T>::load(base + (row + 2 * R) * ld, v2);
    Vec8<T>::load(base + (row + 3 * R) * ld, v3);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (v0[k] + v1[k]) + (v2[k] + v3[k]);
  }
  for (; row < r1; row += R) {
    float v[8];
    Vec8<T>::load(base + row * ld, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl * cols + q * 8 + k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[r * cols + c];
    partial[(int64_t)blockIdx.x * cols + c] = s;
  }
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial, int nparts, int cols,
                                                           float* __restrict__ out) {
  // a block owns 16 columns; 16 threads per column stride over the partial rows, then meet in LDS
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int p = sub;
    for (; p + 16 < nparts; p += 32) {
      s0 += partial[(int64_t)p * cols + c];
      s1 += partial[(int64_t)(p + 16) * cols + c];
    }
    if (p < nparts) s0 += partial[(int64_t)p * cols + c];
  }
  red[sub][cl] = s0 + s1;
  __syncthreads();
  if (sub == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][cl];
    out[c] = s;
  }
}

static int colsum_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;                           // >= 256 rows per block
  if (b > kColsumMaxBlocks) b = kColsumMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace
}  // namespace segger

using namespace segger;

#define DISPATCH_DTYPE(dtype, EXPR_F32, EXPR_BF16, EXPR_F16)           \
  switch (dtype) {                                                      \
    case SEGGER_F32: EXPR_F32; break;                                   \
    case SEGGER_BF16: EXPR_BF16; break;                                 \
    case SEGGER_F16: EXPR_F16; break;                                   \
    default: set_error("unknown dtype %d", dtype); return SEGGER_EINVAL; \
  }

extern "C" int segger_posfreq(const float* pos, const int64_t* batch, const float* mins, const float* maxs, int64_t n,
                              int32_t freq_dim, float eps, float max_period, void* out, int32_t dtype, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && freq_dim >= 16 && freq_dim % 16 == 0, "segger_posfreq: freq_dim must be a positive multiple of 16");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(pos && mins && maxs && out, "segger_posfreq: NULL pointer");
  SEGGER_REQUIRE(aligned16(out), "segger_posfreq: out must be 16-byte aligned");
  const int half = freq_dim / 2;
  const int64_t items = 2 * n * (half / 8);
  const int64_t nb = (items + 255) / 256;
  SEGGER_REQUIRE(nb < 0x7fffffffLL, "segger_posfreq: too many rows");
  const float lmp = logf(max_period);
#define GO(T) hipLaunchKernelGGL((posfreq_kernel<T>), dim3((unsigned)nb), dim3(256), half * sizeof(float), stream, pos, batch, mins, maxs, n, freq_dim, eps, lmp, (T*)out)
  DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  SEGGER_LAUNCH_CHECK("posfreq_kernel");
  return SEGGER_OK;
}

extern "C" int segger_embed_gelu_fwd(const float* table, const int32_t* ids, const void* pe, int64_t ld_pe, int64_t n,
                                     int32_t n_rows_table, int32_t D, void* out, int64_t ld_out, int32_t dtype,
                                     segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && D > 0 && D % 8 == 0, "segger_embed_gelu_fwd: D must be a positive multiple of 8");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(table && ids && pe && out, "segger_embed_gelu_fwd: NULL pointer");
  SEGGER_REQUIRE(aligned16(table) && aligned16(pe) && aligned16(out), "segger_embed_gelu_fwd: 16-byte alignment required");
  SEGGER_REQUIRE((ld_pe * esize(dtype)) % 16 == 0 && (ld_out * esize(dtype)) % 16 == 0 && ld_pe >= D && ld_out >= 2 * D,
                 "segger_embed_gelu_fwd: bad leading dimension");
  const int64_t items = n * (2 * D / 8);
  const int64_t nb = (items + 255) / 256;
  SEGGER_REQUIRE(nb < 0x7fffffffLL, "segger_embed_gelu_fwd: too many rows");
#define GO(T) hipLaunchKernelGGL((embed_gelu_fwd_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, table, ids, (const T*)pe, ld_pe, n, D, (T*)out, ld_out)
  DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  SEGGER_LAUNCH_CHECK("embed_gelu_fwd_kernel");
  return SEGGER_OK;
}

static int64_t emb_max_chunks(int64_t n, int32_t G) {
  const int64_t by_rows = (n + kEmbChunk - 1) / kEmbChunk + G;          // every gene adds at most one ragged chunk
  const int64_t by_genes = (int64_t)G * kEmbMaxChunks;
  return by_rows < by_genes ? by_rows : by_genes;
}

extern "C" size_t segger_embed_gelu_bwd_workspace_bytes(int64_t n, int32_t n_rows_table, int32_t D) {
  if (n < 0 || n_rows_table <= 0 || D <= 0) return 16;
  return (size_t)emb_max_chunks(n, n_rows_table) * (size_t)D * sizeof(float) + ((size_t)n_rows_table + 1) * sizeof(int) + 32;
}

// gtable = gelu'(table) * (rows of gx0[:, :D] summed per id): plan + gather + finish (see emb_gather_kernel)
static int embed_table_grad(const void* gx0, int64_t ld_g, const float* table, int64_t n, int32_t n_rows_table, int32_t D,
                            float* gtable, const int64_t* gene_ptr, const int32_t* gene_rows, void* workspace,
                            size_t workspace_bytes, int32_t dtype, hipStream_t stream) {
  SEGGER_REQUIRE(table && gene_ptr && gene_rows, "embedding-table gradient: table / gene_ptr / gene_rows required");
  const size_t need = segger_embed_gelu_bwd_workspace_bytes(n, n_rows_table, D);
  if (!workspace || workspace_bytes < need) {
    set_error("embedding-table gradient: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t gd = (int64_t)n_rows_table * D;
  const int64_t max_chunks = emb_max_chunks(n, n_rows_table);
  SEGGER_REQUIRE(max_chunks < 0x7fffffffLL, "embedding-table gradient: too many rows");
  float* partial = static_cast<float*>(workspace);
  int* chunk_ptr = reinterpret_cast<int*>(static_cast<char*>(workspace) + (((size_t)max_chunks * D * sizeof(float) + 15) & ~(size_t)15));
  hipLaunchKernelGGL(emb_plan_kernel, dim3(1), dim3(256), 0, stream, gene_ptr, (int)n_rows_table, chunk_ptr);
  const int P = D / 8, R = 256 / P > 0 ? 256 / P : 1;
  const size_t lds = (size_t)R * D * sizeof(float);
#define GO(T) hipLaunchKernelGGL((emb_gather_kernel<T>), dim3((unsigned)max_chunks), dim3((unsigned)(P * R)), lds, stream, (const T*)gx0, ld_g, gene_ptr, gene_rows, chunk_ptr, (int)n_rows_table, D, partial)
  DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  hipLaunchKernelGGL(emb_finish_kernel, dim3((unsigned)((gd + 255) / 256)), dim3(256), 0, stream, partial, chunk_ptr, table, (int)n_rows_table, D, gtable);
  return SEGGER_OK;
}

extern "C" int segger_embed_gelu_bwd(const void* gx0, int64_t ld_g, const float* table, const void* pe, int64_t ld_pe,
                                     int64_t n, int32_t n_rows_table, int32_t D, void* gpe, int64_t ld_gpe, float* gtable,
                                     const int64_t* gene_ptr, const int32_t* gene_rows, void* workspace,
                                     size_t workspace_bytes, int32_t dtype, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && D > 0 && D % 8 == 0 && D <= 2048, "segger_embed_gelu_bwd: D must be a multiple of 8, <= 2048");
  SEGGER_REQUIRE(n_rows_table > 0, "segger_embed_gelu_bwd: empty table");
  const int64_t gd = (int64_t)n_rows_table * D;
  if (n == 0) {
    if (gtable) SEGGER_HIP(hipMemsetAsync(gtable, 0, gd * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(gx0 && pe && gpe, "segger_embed_gelu_bwd: NULL pointer");
  SEGGER_REQUIRE(aligned16(gx0) && aligned16(pe) && aligned16(gpe), "segger_embed_gelu_bwd: 16-byte alignment required");
  SEGGER_REQUIRE((ld_g * esize(dtype)) % 16 == 0 && ld_g >= 2 * D, "segger_embed_gelu_bwd: bad leading dimension of gx0");
  {
    const int64_t items = n * (D / 8);
    const int64_t nb = (items + 255) / 256;
    SEGGER_REQUIRE(nb < 0x7fffffffLL, "segger_embed_gelu_bwd: too many rows");
#define GO(T) hipLaunchKernelGGL((embed_gelu_bwd_pe_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, (const T*)gx0, ld_g, (const T*)pe, ld_pe, n, D, (T*)gpe, ld_gpe)
    DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  }
  if (gtable) {
    const int rc = embed_table_grad(gx0, ld_g, table, n, n_rows_table, D, gtable, gene_ptr, gene_rows, workspace, workspace_bytes,
                                    dtype, stream);
    if (rc != SEGGER_OK) return rc;
  }
  SEGGER_LAUNCH_CHECK("embed_gelu_bwd kernels");
  return SEGGER_OK;
}

static int front_join_fill(const segger_front_join_args* a, bool bwd, FrontJoin* out) {
  SEGGER_REQUIRE(a != nullptr, "segger_front_join: args is NULL");
  SEGGER_REQUIRE(a->n_tx >= 0 && a->n_bd >= 0 && a->D > 0 && a->D % 8 == 0 && a->D <= 2048 && a->n_rows_table > 0,
                 "segger_front_join: sizes (D a multiple of 8, <= 2048)");
  SEGGER_REQUIRE(a->dtype == SEGGER_F32 || a->dtype == SEGGER_BF16 || a->dtype == SEGGER_F16, "segger_front_join: unknown dtype %d", a->dtype);
  const size_t es = esize(a->dtype);
  const int D = a->D;
  auto ok = [&](const void* p, int64_t ld, int width) { return p && aligned16(p) && ld >= width && ((size_t)ld * es) % 16 == 0; };
  const int64_t n_all = a->n_tx + a->n_bd;
  SEGGER_REQUIRE(n_all == 0 || ok(a->pe, a->ld_pe, D), "segger_front_join: pe NULL, misaligned or ld < D");
  SEGGER_REQUIRE(a->n_tx == 0 || (a->table && a->ids && aligned16(a->table)), "segger_front_join: table / ids");
  SEGGER_REQUIRE(a->n_bd == 0 || ok(a->xb, a->ld_xb, D), "segger_front_join: xb NULL, misaligned or ld < D");
  if (!bwd) {
    SEGGER_REQUIRE(a->n_tx == 0 || ok(a->out_tx, a->ld_out_tx, 2 * D), "segger_front_join_fwd: out_tx");
    SEGGER_REQUIRE(a->n_bd == 0 || ok(a->out_bd, a->ld_out_bd, 2 * D), "segger_front_join_fwd: out_bd");
  } else {
    SEGGER_REQUIRE(a->n_tx == 0 || ok(a->g_tx, a->ld_g_tx, 2 * D), "segger_front_join_bwd: g_tx");
    SEGGER_REQUIRE(a->n_bd == 0 || (ok(a->g_bd, a->ld_g_bd, 2 * D) && ok(a->g_xb, a->ld_g_xb, D)), "segger_front_join_bwd: g_bd / g_xb");
    SEGGER_REQUIRE(n_all == 0 || ok(a->g_pe, a->ld_g_pe, D), "segger_front_join_bwd: g_pe");
  }
  FrontJoin p{};
  p.table = a->table; p.ids = a->ids; p.D = D;
  p.pe = a->pe; p.ld_pe = a->ld_pe; p.n_tx = a->n_tx; p.n_bd = a->n_bd;
  p.xb = a->xb; p.ld_xb = a->ld_xb;
  p.out_tx = a->out_tx; p.ld_out_tx = a->ld_out_tx; p.out_bd = a->out_bd; p.ld_out_bd = a->ld_out_bd;
  p.g_tx = a->g_tx; p.ld_g_tx = a->ld_g_tx; p.g_bd = a->g_bd; p.ld_g_bd = a->ld_g_bd;
  p.g_pe = a->g_pe; p.ld_g_pe = a->ld_g_pe; p.g_xb = a->g_xb; p.ld_g_xb = a->ld_g_xb;
  *out = p;
  return SEGGER_OK;
}

extern "C" int segger_front_join_fwd(const segger_front_join_args* a, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FrontJoin p;
  const int rc = front_join_fill(a, false, &p);
  if (rc != SEGGER_OK) return rc;
  const int64_t items = (p.n_tx + p.n_bd) * (2 * p.D / 8);
  if (items == 0) return SEGGER_OK;
  const int64_t nb = (items + 255) / 256;
  SEGGER_REQUIRE(nb < 0x7fffffffLL, "segger_front_join_fwd: too many rows");
#define GO(T) hipLaunchKernelGGL((front_join_fwd_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, p)
  DISPATCH_DTYPE(a->dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  SEGGER_LAUNCH_CHECK("front_join_fwd_kernel");
  return SEGGER_OK;
}

extern "C" int segger_front_join_bwd(const segger_front_join_args* a, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FrontJoin p;
  const int rc = front_join_fill(a, true, &p);
  if (rc != SEGGER_OK) return rc;
  const int64_t items = (p.n_tx + 2 * p.n_bd) * (p.D / 8);
  if (items > 0) {
    const int64_t nb = (items + 255) / 256;
    SEGGER_REQUIRE(nb < 0x7fffffffLL, "segger_front_join_bwd: too many rows");
#define GO(T) hipLaunchKernelGGL((front_join_bwd_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, p)
    DISPATCH_DTYPE(a->dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
    SEGGER_LAUNCH_CHECK("front_join_bwd_kernel");
  }
  if (a->g_table) {
    if (a->n_tx == 0) {
      SEGGER_HIP(hipMemsetAsync(a->g_table, 0, (size_t)a->n_rows_table * a->D * sizeof(float), stream));
      return SEGGER_OK;
    }
    const int rc2 = embed_table_grad(a->g_tx, a->ld_g_tx, a->table, a->n_tx, a->n_rows_table, a->D, a->g_table, a->gene_ptr,
                                     a->gene_rows, a->workspace, a->workspace_bytes, a->dtype, stream);
    if (rc2 != SEGGER_OK) return rc2;
    SEGGER_LAUNCH_CHECK("front_join table gradient");
  }
  return SEGGER_OK;
}

extern "C" size_t segger_segment_rowsum_workspace_bytes(int64_t n, int32_t n_segments, int32_t D) {
  return segger_embed_gelu_bwd_workspace_bytes(n, n_segments, D);
}

extern "C" int segger_segment_rowsum(const void* x, int64_t ld, int64_t n, int32_t D, int32_t dtype, const int64_t* seg_ptr,
                                     const int32_t* seg_rows, int32_t n_segments, float* out, void* workspace,
                                     size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && n_segments > 0 && D > 0 && D % 8 == 0 && D <= 2048, "segger_segment_rowsum: D must be a multiple of 8, <= 2048");
  SEGGER_REQUIRE(out, "segger_segment_rowsum: NULL output");
  const int64_t gd = (int64_t)n_segments * D;
  if (n == 0) {
    SEGGER_HIP(hipMemsetAsync(out, 0, gd * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(x && seg_ptr && seg_rows, "segger_segment_rowsum: NULL pointer");
  SEGGER_REQUIRE(aligned16(x) && ld >= D && (ld * esize(dtype)) % 16 == 0, "segger_segment_rowsum: bad pointer / leading dimension");
  const size_t need = segger_segment_rowsum_workspace_bytes(n, n_segments, D);
  if (!workspace || workspace_bytes < need) {
    set_error("segger_segment_rowsum: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t max_chunks = emb_max_chunks(n, n_segments);
  SEGGER_REQUIRE(max_chunks < 0x7fffffffLL, "segger_segment_rowsum: too many rows");
  float* partial = static_cast<float*>(workspace);
  int* chunk_ptr = reinterpret_cast<int*>(static_cast<char*>(workspace) + (((size_t)max_chunks * D * sizeof(float) + 15) & ~(size_t)15));
  hipLaunchKernelGGL(emb_plan_kernel, dim3(1), dim3(256), 0, stream, seg_ptr, (int)n_segments, chunk_ptr);
  const int P = D / 8, R = 256 / P > 0 ? 256 / P : 1;
  const size_t lds = (size_t)R * D * sizeof(float);
#define GO(T) hipLaunchKernelGGL((emb_gather_kernel<T>), dim3((unsigned)max_chunks), dim3((unsigned)(P * R)), lds, stream, (const T*)x, ld, seg_ptr, seg_rows, chunk_ptr, (int)n_segments, D, partial)
  DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  hipLaunchKernelGGL(emb_finish_kernel, dim3((unsigned)((gd + 255) / 256)), dim3(256), 0, stream, partial, chunk_ptr,
                     (const float*)nullptr, (int)n_segments, D, out);
  SEGGER_LAUNCH_CHECK("segment_rowsum kernels");
  return SEGGER_OK;
}

extern "C" size_t segger_colsum_workspace_bytes(int64_t n, int32_t cols) {
  return (size_t)colsum_blocks(n) * (size_t)(cols > 0 ? cols : 0) * sizeof(float) + 16;
}

extern "C" int segger_colsum(const void* x, int64_t ld, int64_t n, int32_t cols, int32_t dtype, float* out,
                             void* workspace, size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && cols > 0 && cols % 8 == 0 && cols <= 2048, "segger_colsum: cols must be a multiple of 8, <= 2048");
  SEGGER_REQUIRE(out, "segger_colsum: NULL output");
  if (n == 0) {
    SEGGER_HIP(hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(x && aligned16(x) && ld >= cols && (ld * esize(dtype)) % 16 == 0, "segger_colsum: bad pointer / leading dimension");
  const size_t need = segger_colsum_workspace_bytes(n, cols);
  if (!workspace || workspace_bytes < need) {
    set_error("segger_colsum: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int nparts = colsum_blocks(n);
  const int64_t rpb = (n + nparts - 1) / nparts;
  const int P = cols / 8, R = 256 / P > 0 ? 256 / P : 1;
  SEGGER_REQUIRE(P <= 256, "segger_colsum: cols must be <= 2048");
  const size_t lds = (size_t)R * cols * sizeof(float);    // <= 256/P * P*8 * 4 = 8 KiB
  float* partial = static_cast<float*>(workspace);
#define GO(T) hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3((unsigned)nparts), dim3((unsigned)(P * R)), lds, stream, (const T*)x, ld, n, cols, rpb, partial)
  DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, stream, partial, nparts, cols, out);
  SEGGER_LAUNCH_CHECK("colsum kernels");
  return SEGGER_OK;
}

extern "C" int segger_l2norm_fwd(const void* y, int64_t ld_y, int64_t n, int32_t channels, float eps, void* z, int64_t ld_z,
                                 int32_t dtype, segger_stream_t stream) {
  SEGGER_REQUIRE(n >= 0 && channels > 0, "segger_l2norm_fwd: bad sizes");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(y && z && aligned16(y) && aligned16(z), "segger_l2norm_fwd: NULL or misaligned pointer");
  SEGGER_REQUIRE((ld_y * esize(dtype)) % 16 == 0 && (ld_z * esize(dtype)) % 16 == 0, "segger_l2norm_fwd: bad leading dimension");
  DISPATCH_DTYPE(dtype,
                 return (launch_l2norm<float, false>(y, ld_y, nullptr, 0, n, channels, eps, z, ld_z, (hipStream_t)stream)),
                 return (launch_l2norm<bf16_t, false>(y, ld_y, nullptr, 0, n, channels, eps, z, ld_z, (hipStream_t)stream)),
                 return (launch_l2norm<f16_t, false>(y, ld_y, nullptr, 0, n, channels, eps, z, ld_z, (hipStream_t)stream)))
  return SEGGER_OK;
}

extern "C" int segger_l2norm_bwd2(const void* y, int64_t ld_y, const void* gz, int64_t ld_gz, const void* gz2, int64_t ld_gz2,
                                  int64_t n, int32_t channels, float eps, void* gy, int64_t ld_gy, int32_t dtype,
                                  segger_stream_t stream) {
  SEGGER_REQUIRE(n >= 0 && channels > 0, "segger_l2norm_bwd: bad sizes");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(y && gz && gy && aligned16(y) && aligned16(gz) && aligned16(gy) && aligned16(gz2),
                 "segger_l2norm_bwd: NULL or misaligned pointer");
  const size_t es = esize(dtype);
  SEGGER_REQUIRE((ld_y * es) % 16 == 0 && (ld_gz * es) % 16 == 0 && (ld_gy * es) % 16 == 0 && (!gz2 || (ld_gz2 * es) % 16 == 0),
                 "segger_l2norm_bwd: bad leading dimension");
  DISPATCH_DTYPE(dtype,
                 return (launch_l2norm<float, true>(y, ld_y, gz, ld_gz, n, channels, eps, gy, ld_gy, (hipStream_t)stream, gz2, ld_gz2)),
                 return (launch_l2norm<bf16_t, true>(y, ld_y, gz, ld_gz, n, channels, eps, gy, ld_gy, (hipStream_t)stream, gz2, ld_gz2)),
                 return (launch_l2norm<f16_t, true>(y, ld_y, gz, ld_gz, n, channels, eps, gy, ld_gy, (hipStream_t)stream, gz2, ld_gz2)))
  return SEGGER_OK;
}

extern "C" int segger_l2norm_bwd(const void* y, int64_t ld_y, const void* gz, int64_t ld_gz, int64_t n, int32_t channels,
                                 float eps, void* gy, int64_t ld_gy, int32_t dtype, segger_stream_t stream) {
  return segger_l2norm_bwd2(y, ld_y, gz, ld_gz, nullptr, 0, n, channels, eps, gy, ld_gy, dtype, stream);
}

extern "C" int segger_l2norm_many(const segger_l2norm_seg* segs, int32_t n_segs, int32_t channels, float eps, int32_t dtype,
                                  segger_stream_t stream) {
  SEGGER_REQUIRE(n_segs >= 0 && n_segs <= kL2MaxSegs && (n_segs == 0 || segs), "segger_l2norm_many: 0..4 segments");
  SEGGER_REQUIRE(channels == 8 || channels == 16 || channels == 32 || channels == 64 || channels == 128,
                 "segger_l2norm_many: channels=%d not supported (8,16,32,64,128)", channels);
  const size_t es = esize(dtype);
  const int lpc = channels / 8;
  const int64_t rpb = 4 * (64 / lpc);
  L2Batch b{};
  b.eps = eps;
  int64_t blocks = 0;
  for (int i = 0; i < n_segs; ++i) {
    const segger_l2norm_seg& g = segs[i];
    SEGGER_REQUIRE(g.n >= 0, "segger_l2norm_many: segment %d: negative size", i);
    if (g.n == 0) continue;
    SEGGER_REQUIRE(g.y && g.out && aligned16(g.y) && aligned16(g.out) && aligned16(g.gz), "segger_l2norm_many: segment %d: NULL or misaligned pointer", i);
    SEGGER_REQUIRE((g.ld_y * es) % 16 == 0 && (g.ld_out * es) % 16 == 0 && g.ld_y >= channels && g.ld_out >= channels &&
                       (!g.gz || ((g.ld_gz * (g.gz_f32 ? 4 : es)) % 16 == 0 && g.ld_gz >= channels)),
                   "segger_l2norm_many: segment %d: bad leading dimension", i);
    b.seg[b.n] = g;
    b.first_block[b.n++] = (int32_t)blocks;
    blocks += (g.n + rpb - 1) / rpb;
    SEGGER_REQUIRE(blocks < 0x7fffffffLL, "segger_l2norm_many: too many rows");
  }
  b.first_block[b.n] = (int32_t)blocks;
  if (blocks == 0) return SEGGER_OK;
#define GO(T, LPC) hipLaunchKernelGGL((l2norm_many_kernel<T, LPC>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b)
#define BY_C(T) switch (lpc) { case 1: GO(T, 1); break; case 2: GO(T, 2); break; case 4: GO(T, 4); break; case 8: GO(T, 8); break; default: GO(T, 16); break; }
  DISPATCH_DTYPE(dtype, BY_C(float), BY_C(bf16_t), BY_C(f16_t))
#undef BY_C
#undef GO
  SEGGER_LAUNCH_CHECK("l2norm_many_kernel");
  return SEGGER_OK;
}
