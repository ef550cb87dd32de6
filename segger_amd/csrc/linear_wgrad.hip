// Weight and bias gradients of a tall-skinny projection on the matrix cores:
//     dW[M, K] = dY[n, M]^T * X[n, K]        db[M] = sum_n dY[n, :]
// for n ~ 10^6 node rows and M, K <= 384 (the lin_l / lin_r / lin_last / positional-MLP maps of the encoder).
// PyG / torch leave this to autograd's `grad.t() @ x` and `grad.sum(0)`: two more passes over dY.
//
// The product reduces over the LONG dimension, so the work is split over row slabs, one per workgroup
// (persistent: <= a few workgroups per CU), and every workgroup keeps the WHOLE [M, K] fp32 accumulator in the
// registers of its 4 or 8 waves (M = 384, K = 256: 96 32x32 tiles, 12 per wave) -- dY and X are read exactly once.
//   * 16 rows per stage: dY[16, M] and X[16, K] are staged row-major into LDS (16-byte pieces, double buffered,
//     next stage's global loads in flight under the MFMAs);
//   * both MFMA operands need the ROW index as their k dimension (A[m][k] = dY[n0+k][m], B[k][c] = X[n0+k][c]),
//     i.e. a transpose of what sits in LDS: ds_read_b64_tr_b16 delivers it for free.  Row stride = width*2 + 64
//     bytes (== 64 or 192 mod 256): the 4 rows of a transposing read fall into 4 disjoint 16-bank ranges;
//   * db comes from the A fragments with v_dot2_f32_bf16 / v_dot2_f32_f16 against a vector of ones (4 instructions
//     per fragment, fp32 accumulate), by the waves that own the first K tiles;
//   * every workgroup writes its partial [M*K + M] to the workspace; a second kernel sums the partials in slab
//     order (deterministic, no atomics).
#include "common.h"

namespace segger {
namespace {

typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v  __attribute__((ext_vector_type(2)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef short    s16x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2  __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

constexpr int kStageRows = 16;           // rows per LDS stage = one 32x32x16 k-step

template <typename T> struct WgMfma;
template <> struct WgMfma<bf16_t> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // c + lo(a) + hi(a)
  static __device__ __forceinline__ float sum2(uint32_t a, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, a), __builtin_bit_cast(bf16x2v, 0x3f803f80u), c, false);
  }
};
template <> struct WgMfma<f16_t> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float sum2(uint32_t a, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2v, a), __builtin_bit_cast(f16x2v, 0x3c003c00u), c, false);
  }
};

// 4 rows x 16 columns of 16-bit elements, transposed: lane i of a 16-lane group receives column i of the 4 rows
__device__ __forceinline__ u32x2 lds_read_tr(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(p)));
  return __builtin_bit_cast(u32x2, v);
}

struct WgradParams {
  const void* dy; int64_t ld_dy;
  const void* x;  int64_t ld_x;
  int64_t n_rows;
  int64_t n_stages;          // ceil(n_rows / 16)
  int64_t stages_per_block;
  float* partial;            // [gridDim.x][M*K + M]
};

template <typename T, int M, int K, int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_kernel(WgradParams p) {
  constexpr int NT = NW * 64;
  constexpr int WM = NW == 8 ? 4 : 2, WK = 2;          // wave grid over (M tiles, K tiles)
  static_assert(M * K / (64 * NW) <= 192, "accumulator does not fit the register file");
  constexpr int TM = M / 32, TK = K / 32;
  static_assert(TM % WM == 0 && TK % WK == 0, "tile grid does not split over the waves");
  constexpr int MT = TM / WM, KT = TK / WK;            // 32x32 tiles per wave
  constexpr int SY = M * 2 + 64, SX = K * 2 + 64;      // LDS row strides in bytes
  constexpr int BUF = kStageRows * (SY + SX);
  constexpr int PY = kStageRows * (M / 8), PX = kStageRows * (K / 8);   // 16-byte pieces per stage
  constexpr int NP = (PY + PX + NT - 1) / NT;          // pieces per thread
  static_assert(PY % 64 == 0, "dY / X piece boundary must be wave aligned");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wk = wave / WM;
  const T* __restrict__ dy = static_cast<const T*>(p.dy);
  const T* __restrict__ x = static_cast<const T*>(p.x);

  // ---- staging: thread -> (matrix, row, 16-byte chunk) for each of its pieces.  Loads go through buffer
  //      resources based at this workgroup's first row: the address of a piece is an SGPR base + a per-thread
  //      32-bit offset fixed for the whole kernel + a scalar stage offset, and rows past the end of the matrix
  //      (the last, partial stage) come back as zeros from the range check -- no branches, no 64-bit VGPR math
  const int64_t s_beg = (int64_t)blockIdx.x * p.stages_per_block;
  int64_t s_end = s_beg + p.stages_per_block;
  if (s_end > p.n_stages) s_end = p.n_stages;
  const int64_t row_beg = s_beg * kStageRows;
  const int64_t rows_here = p.n_rows > row_beg ? p.n_rows - row_beg : 0;       // valid rows from row_beg on
  const int64_t span_rows = rows_here < p.stages_per_block * kStageRows ? rows_here : p.stages_per_block * kStageRows;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(dy) + row_beg * p.ld_dy, 0, (int)(span_rows * p.ld_dy * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(x) + row_beg * p.ld_x, 0, (int)(span_rows * p.ld_x * 2), 0x00020000);
  const int stage_bytes_y = kStageRows * (int)p.ld_dy * 2, stage_bytes_x = kStageRows * (int)p.ld_x * 2;
  int voff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + i * NT;
    if (q < PY) voff[i] = (q / (M / 8)) * (int)p.ld_dy * 2 + (q % (M / 8)) * 16;
    else        voff[i] = ((q - PY) / (K / 8)) * (int)p.ld_x * 2 + ((q - PY) % (K / 8)) * 16;
  }
  u32x4 st[NP];
  auto fetch = [&](int local_stage) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * NT;                      // wave-uniform choice: PY is a multiple of 64
      if (q < PY)
        st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, voff[i], local_stage * stage_bytes_y, 0));
      else if (q < PY + PX)
        st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff[i], local_stage * stage_bytes_x, 0));
    }
  };
  auto commit = [&](int buf) {
    unsigned char* by = lds + buf * BUF;
    unsigned char* bx = by + kStageRows * SY;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = tid + i * NT;
      if (q < PY) {
        const int row = q / (M / 8), ch = q % (M / 8);
        *reinterpret_cast<u32x4*>(by + row * SY + ch * 16) = st[i];
      } else if (q < PY + PX) {
        const int qq = q - PY;
        const int row = qq / (K / 8), ch = qq % (K / 8);
        *reinterpret_cast<u32x4*>(bx + row * SX + ch * 16) = st[i];
      }
    }
  };

  // ---- transposing-read addresses of this lane (see the header comment) -------------------------------------
  const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const int row_a = 8 * (g >> 1) + tq;                 // + 4 for the second half of the fragment
  const int col_a = 16 * (g & 1) + 4 * tp;             // + 32 * tile
  const int off_y = row_a * SY + col_a * 2 + wm * MT * 64;
  const int off_x = kStageRows * SY + row_a * SX + col_a * 2 + wk * KT * 64;

  f32x16 acc[MT][KT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  float dbias[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) dbias[a] = 0.f;

  const int n_local = (int)(s_end > s_beg ? s_end - s_beg : 0);
  if (n_local > 0) {
    fetch(0);
    commit(0);
  }
  __syncthreads();
  for (int s = 0; s < n_local; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < n_local;
    if (more) fetch(s + 1);                            // global loads in flight under the MFMAs
    const unsigned char* base = lds + buf * BUF;
    u32x4 fa[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const u32x2 lo = lds_read_tr(base + off_y + a * 64);
      const u32x2 hi = lds_read_tr(base + off_y + a * 64 + 4 * SY);
      fa[a] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    if (wk == 0) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        float d = dbias[a];
        d = WgMfma<T>::sum2(fa[a].x, d); d = WgMfma<T>::sum2(fa[a].y, d);
        d = WgMfma<T>::sum2(fa[a].z, d); d = WgMfma<T>::sum2(fa[a].w, d);
        dbias[a] = d;
      }
    }
#pragma unroll
    for (int b = 0; b < KT; ++b) {
      const u32x2 lo = lds_read_tr(base + off_x + b * 64);
      const u32x2 hi = lds_read_tr(base + off_x + b * 64 + 4 * SX);
      const u32x4 fb = u32x4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
      for (int a = 0; a < MT; ++a) acc[a][b] = WgMfma<T>::run(fa[a], fb, acc[a][b]);
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- partial results: acc tile (a, b) element e of lane l is dW[m][k] with
  //      m = 32*(wm*MT + a) + (e & 3) + 8*(e >> 2) + 4*(l >> 5),  k = 32*(wk*KT + b) + (l & 31) -----------------------
  float* out = p.partial + (int64_t)blockIdx.x * (M * K + M);
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
#pragma unroll
    for (int b = 0; b < KT; ++b) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = 32 * (wm * MT + a) + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int k = 32 * (wk * KT + b) + r;
        out[m * K + k] = acc[a][b][e];
      }
    }
  }
  if (wk == 0) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const float d = dbias[a] + __shfl_xor(dbias[a], 32, 64);      // rows 8h + j of both halves
      if (h == 0) out[M * K + 32 * (wm * MT + a) + r] = d;
    }
  }
}

// out[e] = sum_s partial[s][e]: grad_w = first M*K entries, grad_b the rest (fixed slab order: deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int64_t n_slabs, int64_t width,
                                                          int64_t mk, float* __restrict__ grad_w, float* __restrict__ grad_b) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= width) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t s = 0;
  for (; s + 4 <= n_slabs; s += 4) {
    s0 += partial[(s + 0) * width + e]; s1 += partial[(s + 1) * width + e];
    s2 += partial[(s + 2) * width + e]; s3 += partial[(s + 3) * width + e];
  }
  for (; s < n_slabs; ++s) s0 += partial[s * width + e];
  const float t = (s0 + s1) + (s2 + s3);
  if (e < mk) grad_w[e] = t;
  else if (grad_b) grad_b[e - mk] = t;
}

constexpr int kNumCu = 256;

bool shape_ok(int m, int k) {
  return (m == 384 || m == 192 || m == 128 || m == 64) && (k == 256 || k == 128 || k == 64);
}
int waves_for(int m) { return m % 128 == 0 ? 8 : 4; }
// workgroups per CU the register / LDS footprint allows (accumulator registers per lane = M*K / (64*NW))
int blocks_per_cu(int m, int k) {
  const int acc = m * k / (64 * waves_for(m));
  if (acc > 96) return 1;
  if (acc > 32) return 2;
  return 4;
}
int64_t grid_for(int64_t n_rows, int m, int k) {
  const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
  const int64_t cap = (int64_t)kNumCu * blocks_per_cu(m, k);
  return stages < cap ? (stages < 1 ? 1 : stages) : cap;
}

template <typename T, int M, int K>
void launch_wgrad(const WgradParams& p, int64_t grid, hipStream_t stream) {
  constexpr int NW = M % 128 == 0 ? 8 : 4;
  hipLaunchKernelGGL((wgrad_kernel<T, M, K, NW>), dim3((unsigned)grid), dim3(NW * 64), 0, stream, p);
}

template <typename T>
int dispatch_wgrad(const WgradParams& p, int m, int k, int64_t grid, hipStream_t stream) {
#define CASE(MM, KK) if (m == MM && k == KK) { launch_wgrad<T, MM, KK>(p, grid, stream); return SEGGER_OK; }
  CASE(384, 256) CASE(384, 128) CASE(384, 64)
  CASE(192, 256) CASE(192, 128) CASE(192, 64)
  CASE(128, 256) CASE(128, 128) CASE(128, 64)
  CASE(64, 256) CASE(64, 128) CASE(64, 64)
#undef CASE
  set_error("segger_linear_wgrad: m_out=%d k_in=%d not supported", m, k);
  return SEGGER_EUNSUPPORTED;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_linear_wgrad_supported(int32_t m_out, int32_t k_in, int32_t dtype) {
  return shape_ok(m_out, k_in) && (dtype == SEGGER_BF16 || dtype == SEGGER_F16);
}

extern "C" size_t segger_linear_wgrad_workspace_bytes(int64_t n_rows, int32_t m_out, int32_t k_in) {
  if (n_rows <= 0 || !shape_ok(m_out, k_in)) return 16;
  return (size_t)grid_for(n_rows, m_out, k_in) * ((size_t)m_out * k_in + m_out) * sizeof(float);
}

extern "C" int segger_linear_wgrad(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, int64_t n_rows,
                                   int32_t m_out, int32_t k_in, int32_t dtype, float* grad_w, float* grad_b,
                                   void* workspace, size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_rows >= 0 && m_out > 0 && k_in > 0, "segger_linear_wgrad: bad sizes");
  SEGGER_REQUIRE(grad_w != nullptr, "segger_linear_wgrad: grad_w is NULL");
  if (!segger_linear_wgrad_supported(m_out, k_in, dtype)) {
    set_error("segger_linear_wgrad: m_out=%d k_in=%d dtype=%d not supported (m_out in {64,128,192,384}, k_in in "
              "{64,128,256}, bf16/f16)", m_out, k_in, dtype);
    return SEGGER_EUNSUPPORTED;
  }
  if (n_rows == 0) {
    SEGGER_HIP(hipMemsetAsync(grad_w, 0, (size_t)m_out * k_in * sizeof(float), stream));
    if (grad_b) SEGGER_HIP(hipMemsetAsync(grad_b, 0, (size_t)m_out * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(dy && x, "segger_linear_wgrad: NULL input");
  SEGGER_REQUIRE(aligned16(dy) && aligned16(x), "segger_linear_wgrad: inputs must be 16-byte aligned");
  SEGGER_REQUIRE(ld_dy >= m_out && ld_x >= k_in && (ld_dy * 2) % 16 == 0 && (ld_x * 2) % 16 == 0,
                 "segger_linear_wgrad: bad leading dimension");
  const size_t need = segger_linear_wgrad_workspace_bytes(n_rows, m_out, k_in);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_linear_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t grid = grid_for(n_rows, m_out, k_in);
  {
    // buffer resources address one workgroup's slab with 32-bit offsets
    const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
    const int64_t span = ((stages + grid - 1) / grid) * kStageRows * (ld_dy > ld_x ? ld_dy : ld_x) * 2;
    SEGGER_REQUIRE(span < 0x7fffffffLL, "segger_linear_wgrad: a workgroup's row slab exceeds 2 GiB");
  }
  WgradParams p{dy, ld_dy, x, ld_x, n_rows, (n_rows + kStageRows - 1) / kStageRows, 0, static_cast<float*>(workspace)};
  p.stages_per_block = (p.n_stages + grid - 1) / grid;
  const int rc = dtype == SEGGER_BF16 ? dispatch_wgrad<bf16_t>(p, m_out, k_in, grid, stream)
                                      : dispatch_wgrad<f16_t>(p, m_out, k_in, grid, stream);
  if (rc != SEGGER_OK) return rc;
  SEGGER_LAUNCH_CHECK("wgrad_kernel");
  const int64_t width = (int64_t)m_out * k_in + m_out;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, stream,
                     p.partial, grid, width, (int64_t)m_out * k_in, grad_w, grad_b);
  SEGGER_LAUNCH_CHECK("wgrad_reduce_kernel");
  return SEGGER_OK;
}
