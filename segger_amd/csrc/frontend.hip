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
  for (int k = 0; k < 8; ++k) sincosf(p * freqs[j0 + k], &sn[k], &cs[k]);
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

// backward, embedding table: gtable[g, c] = sum_{n: ids[n]=g} gx0[n, c] * gelu'(table[g, c]).
// Block (bx, by) owns a 32-channel slice `by` and a contiguous row range `bx`; it sums
// sum_n gx0[n, c] per gene in an LDS table [G][32] (ds_add_f32), then writes the table to
// partial[bx][G][D]; gelu'(table) is applied once per table entry by the reduce kernel.
constexpr int kEmbSlice = 32;      // D must be a multiple of this
// Block (bx, by) owns the channel slice [by*S, by*S+S) (S = 64 when the LDS table fits, else 32) and a
// contiguous row range; a thread loads 8 channels of one row (16 B, a full 128-B line per row at S=64)
// and adds them into the block's LDS table [G][S+1] (the +1 skews genes over banks).
template <typename T>
__global__ __launch_bounds__(256) void embed_grad_partial_kernel(const T* __restrict__ gx0, int64_t ld_g, const int32_t* __restrict__ ids,
                                                                int64_t n, int G, int D, int S, int64_t rows_per_block,
                                                                float* __restrict__ partial) {
  extern __shared__ float tab[];                         // [G][S + 1]
  const int stride = S + 1;
  for (int i = threadIdx.x; i < G * stride; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int tpr = S / 8;                                 // threads per row
  const int c8 = (threadIdx.x % tpr) * 8;
  const int rl = threadIdx.x / tpr;
  const int c0 = blockIdx.y * S;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int64_t row = r0 + rl; row < r1; row += 256 / tpr) {
    const int g = ids[row];
    float v[8];
    Vec8<T>::load(gx0 + row * ld_g + c0 + c8, v);
    float* t = tab + g * stride + c8;
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(t + k, v[k]);  // LDS atomics
  }
  __syncthreads();
  float* dst = partial + (int64_t)blockIdx.x * G * D;
  for (int i = threadIdx.x; i < G * S; i += 256) {
    const int g = i / S, cc = i % S;
    dst[(int64_t)g * D + c0 + cc] = tab[g * stride + cc];
  }
}

__global__ __launch_bounds__(256) void embed_grad_reduce_kernel(const float* __restrict__ partial, int nparts, const float* __restrict__ table,
                                                               int64_t gd, float* __restrict__ gtable) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= gd) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += partial[(int64_t)p * gd + i];
  gtable[i] = s * gelu_erf_grad(table[i]);
}

// ---------------------------------------------------------------------------------------------
// L2 row normalisation: z = y / max(||y||, eps)  ;  gy = (gz - z * <z, gz>) / max(||y||, eps)
// (exact for ||y|| >= eps; below eps torch's clamp makes the op linear: gy = gz / eps)
// LPC = C/8 lanes per row.
// ---------------------------------------------------------------------------------------------
template <typename T, int LPC, bool BWD>
__global__ __launch_bounds__(256) void l2norm_kernel(const T* __restrict__ y, int64_t ld_y, const T* __restrict__ gz, int64_t ld_gz,
                                                    int64_t n, float eps, T* __restrict__ out, int64_t ld_out) {
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
                  int64_t ld_out, hipStream_t stream) {
#define GO(LPC)                                                                                                      \
  {                                                                                                                  \
    const int64_t rpb = 4 * (64 / LPC);                                                                              \
    hipLaunchKernelGGL((l2norm_kernel<T, LPC, BWD>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, stream,    \
                       (const T*)y, ld_y, (const T*)gz, ld_gz, n, eps, (T*)out, ld_out);                             \
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

size_t esize(int dtype) { return dtype == SEGGER_F32 ? 4 : 2; }

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

static int emb_row_blocks(int64_t n) {
  int64_t b = (n + 4095) / 4096;          // >= 4096 rows per block
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t segger_embed_gelu_bwd_workspace_bytes(int64_t n, int32_t n_rows_table, int32_t D) {
  return (size_t)emb_row_blocks(n) * (size_t)n_rows_table * (size_t)D * sizeof(float) + 16;
}

extern "C" int segger_embed_gelu_bwd(const void* gx0, int64_t ld_g, const float* table, const int32_t* ids, const void* pe,
                                     int64_t ld_pe, int64_t n, int32_t n_rows_table, int32_t D, void* gpe, int64_t ld_gpe,
                                     float* gtable, void* workspace, size_t workspace_bytes, int32_t dtype,
                                     segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && D > 0 && D % kEmbSlice == 0, "segger_embed_gelu_bwd: D must be a positive multiple of 32");
  SEGGER_REQUIRE(n_rows_table > 0, "segger_embed_gelu_bwd: empty table");
  const int64_t gd = (int64_t)n_rows_table * D;
  if (n == 0) {
    if (gtable) SEGGER_HIP(hipMemsetAsync(gtable, 0, gd * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(gx0 && pe && gpe, "segger_embed_gelu_bwd: NULL pointer");
  SEGGER_REQUIRE(aligned16(gx0) && aligned16(pe) && aligned16(gpe), "segger_embed_gelu_bwd: 16-byte alignment required");
  // widest channel slice whose [G][S+1] fp32 table fits the 160 KiB LDS
  // (64-wide slices only while the table stays within the default 64 KiB dynamic-LDS limit: larger requests
  //  need hipFuncSetAttribute, and hipGraph instantiation of such kernel nodes crashed on ROCm 7.0/7.2)
  const int S = (D % 64 == 0 && (size_t)n_rows_table * 65 * sizeof(float) <= 64 * 1024) ? 64 : kEmbSlice;
  const size_t lds_bytes = (size_t)n_rows_table * (S + 1) * sizeof(float);
  {
    const int64_t items = n * (D / 8);
    const int64_t nb = (items + 255) / 256;
#define GO(T) hipLaunchKernelGGL((embed_gelu_bwd_pe_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, (const T*)gx0, ld_g, (const T*)pe, ld_pe, n, D, (T*)gpe, ld_gpe)
    DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
  }
  if (gtable) {
    SEGGER_REQUIRE(table && ids, "segger_embed_gelu_bwd: table / ids required for the table gradient");
    if (lds_bytes > 160 * 1024) {
      set_error("segger_embed_gelu_bwd: %d table rows need %zu B of LDS (> 160 KiB)", n_rows_table, lds_bytes);
      return SEGGER_EUNSUPPORTED;
    }
    const int nparts = emb_row_blocks(n);
    const size_t need = segger_embed_gelu_bwd_workspace_bytes(n, n_rows_table, D);
    if (!workspace || workspace_bytes < need) {
      set_error("segger_embed_gelu_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
      return SEGGER_EWORKSPACE;
    }
    const int64_t rpb = (n + nparts - 1) / nparts;
    float* partial = static_cast<float*>(workspace);
    dim3 grid((unsigned)nparts, (unsigned)(D / S));
#define GO(T)                                                                                                         \
  do {                                                                                                                \
    static bool attr_set = false;   /* once per type: not a stream operation, keep it out of graph captures */   \
    if (!attr_set) {                                                                                                  \
      (void)hipFuncSetAttribute((const void*)embed_grad_partial_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((embed_grad_partial_kernel<T>), grid, dim3(256), lds_bytes, stream, (const T*)gx0, ld_g, ids, n, n_rows_table, D, S, rpb, partial); \
  } while (0)
    DISPATCH_DTYPE(dtype, GO(float), GO(bf16_t), GO(f16_t))
#undef GO
    hipLaunchKernelGGL(embed_grad_reduce_kernel, dim3((unsigned)((gd + 255) / 256)), dim3(256), 0, stream, partial, nparts, table, gd, gtable);
  }
  SEGGER_LAUNCH_CHECK("embed_gelu_bwd kernels");
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

extern "C" int segger_l2norm_bwd(const void* y, int64_t ld_y, const void* gz, int64_t ld_gz, int64_t n, int32_t channels,
                                 float eps, void* gy, int64_t ld_gy, int32_t dtype, segger_stream_t stream) {
  SEGGER_REQUIRE(n >= 0 && channels > 0, "segger_l2norm_bwd: bad sizes");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(y && gz && gy && aligned16(y) && aligned16(gz) && aligned16(gy), "segger_l2norm_bwd: NULL or misaligned pointer");
  DISPATCH_DTYPE(dtype,
                 return (launch_l2norm<float, true>(y, ld_y, gz, ld_gz, n, channels, eps, gy, ld_gy, (hipStream_t)stream)),
                 return (launch_l2norm<bf16_t, true>(y, ld_y, gz, ld_gz, n, channels, eps, gy, ld_gy, (hipStream_t)stream)),
                 return (launch_l2norm<f16_t, true>(y, ld_y, gz, ld_gz, n, channels, eps, gy, ld_gy, (hipStream_t)stream)))
  return SEGGER_OK;
}
