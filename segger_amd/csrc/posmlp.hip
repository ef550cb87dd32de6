// The positional embedder as ONE kernel (reference src/segger/models/ist_encoder.py:33-79):
//   per coordinate row (node n, axis c):  p = (pos - lo) / (hi - lo + eps)          per-graph min-max normalisation
//   F  = [cos(p w_j) | sin(p w_j)], j < 128                                         sinusoidal_embedding, 256 wide
//   z1 = F W0^T + b0 (64),  h1 = SiLU(z1),  pe = h1 W2^T + b2 (64)                  the shared MLP
//
// The unfused route writes F -- [2 Nt, 256], 1 GB at C2 -- to HBM and reads it back for the first GEMM.  Here a lane
// GENERATES its rows' features directly in the register layout of the MFMA B operand (lane (r, h) of a
// v_mfma_f32_32x32x16 supplies 8 consecutive k for data row r: 8 frequencies of one coordinate), so F never exists in
// memory; for training the kernel stores the normalised coordinate (4 bytes per row) and the weight gradient
// dW0 = dz1^T F regenerates F the same way (segger_posmlp_wgrad: linear_wgrad.hip's kernel with a generated operand).  The second GEMM consumes the first one's accumulators without a transpose:
// the k order of a GEMM is free, so the A operand (W2) is read from LDS in the order the accumulator lanes hold h1.
#include "common.h"

namespace segger {
namespace {

typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma<f16_t> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

constexpr int kFreq = 256, kHalf = 128, kDim = 64;
constexpr int kW0Stride = kFreq * 2 + 16;      // bytes per LDS row of W0 [64, 256]
constexpr int kW2Stride = kDim * 2 + 16;       // ... of W2 [64, 64]
constexpr int kEStride = kDim * 2 + 16;        // ... of a wave's 32 x 64 epilogue tile

struct PosMlpParams {
  const float* pos; const int64_t* batch; const float* mins; const float* maxs;
  int64_t n; float eps; float log_max_period;
  const void* w0; const float* b0; const void* w2; const float* b2;
  void* pe; void* z1; float* pn;
  void* pe_pre;            // GELU && TRAIN: the embedder's own output; `pe` then receives gelu(pe_pre)
  void* h1;                // TRAIN, optional: SiLU(z1), the second layer's input (for the unfused weight gradient)
};

// sin / cos on the hardware units (v_sin_f32 / v_cos_f32 take revolutions and reduce the range themselves; absolute
// error ~1e-6, two orders below the rounding of the 16-bit operand they feed): 3 instructions per angle.  libm's
// sincosf is ~70 (its Payne-Hanek path gets if-converted into the loop) and made this kernel 3x slower than the
// memory traffic it saves.
__device__ __forceinline__ void sincos_hw(float x, float* s, float* c) {
  const float rev = x * 0.15915494309189535f;
  *s = __builtin_amdgcn_sinf(rev);
  *c = __builtin_amdgcn_cosf(rev);
}

template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  return u32x4{Vec8<T>::pack(v[0], v[1]), Vec8<T>::pack(v[2], v[3]), Vec8<T>::pack(v[4], v[5]), Vec8<T>::pack(v[6], v[7])};
}

// the wave's 32 x 64 tile `et` (lane (r, h) wrote its 4-column groups) -> 32 rows of 128 bytes in memory
template <typename T>
__device__ __forceinline__ void store_tile(const unsigned char* et, T* dst, int64_t row0, int64_t n_rows, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int er = 8 * i + (lane >> 3), piece = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(et + er * kEStride + piece * 16);
    const int64_t row = row0 + er;
    if (row < n_rows) *reinterpret_cast<u32x4*>(dst + row * kDim + piece * 8) = v;
  }
}

// 8 waves per workgroup, 2 workgroups per CU (4 waves per SIMD, <= 128 registers per lane): the per-tile work is a
// dependent chain (coordinate -> sin / cos -> MFMA -> SiLU -> MFMA -> GELU -> stores) that needs the other waves to fill
// the matrix and transcendental pipes; the biases live in LDS (16 floats per tile and lane) instead of 64 registers
constexpr int kPmWaves = 8;
template <typename T, bool TRAIN, bool GELU>
__global__ __launch_bounds__(kPmWaves * 64, 2) void posmlp_fwd_kernel(PosMlpParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kDim * kW0Stride + kDim * kW2Stride + kPmWaves * 32 * kEStride +
                                                            kHalf * 4 + 2 * kDim * 4];
  unsigned char* lw0 = lds;
  unsigned char* lw2 = lds + kDim * kW0Stride;
  unsigned char* le = lw2 + kDim * kW2Stride;
  float* freqs = reinterpret_cast<float*>(le + kPmWaves * 32 * kEStride);
  float* lb0 = freqs + kHalf;
  float* lb2 = lb0 + kDim;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  {
    const T* w0 = static_cast<const T*>(p.w0);
    const T* w2 = static_cast<const T*>(p.w2);
    for (int piece = tid; piece < kDim * (kFreq / 8); piece += kPmWaves * 64) {
      const int row = piece / (kFreq / 8), col = piece % (kFreq / 8);
      *reinterpret_cast<u32x4*>(lw0 + row * kW0Stride + col * 16) = *reinterpret_cast<const u32x4*>(w0 + row * kFreq + col * 8);
    }
    for (int piece = tid; piece < kDim * (kDim / 8); piece += kPmWaves * 64) {
      const int row = piece / (kDim / 8), col = piece % (kDim / 8);
      *reinterpret_cast<u32x4*>(lw2 + row * kW2Stride + col * 16) = *reinterpret_cast<const u32x4*>(w2 + row * kDim + col * 8);
    }
    if (tid < kHalf) freqs[tid] = expf(-p.log_max_period * (float)tid / (float)kHalf);
    if (tid < kDim) { lb0[tid] = p.b0[tid]; lb2[tid] = p.b2[tid]; }
  }
  __syncthreads();

  unsigned char* et = le + wave * 32 * kEStride;
  const int64_t n_rows = 2 * p.n;
  const int64_t n_tiles = (n_rows + 31) / 32;
  T* pe = static_cast<T*>(p.pe);
  // normalised coordinate of this lane's row in tile t (a chain of dependent loads: batch id -> per-graph min / max);
  // the next tile's is requested before this tile's arithmetic starts
  auto coord = [&](int64_t t, bool& ok, int64_t& row_out) -> float {
    ok = t * 32 + r < n_rows;
    const int64_t row = ok ? t * 32 + r : n_rows - 1;
    row_out = row;
    const int64_t node = row >> 1;
    const int c = (int)(row & 1);
    const int64_t g = p.batch ? p.batch[node] : 0;
    const float lo = p.mins[2 * g + c], hi = p.maxs[2 * g + c];
    return (p.pos[2 * node + c] - lo) / (hi - lo + p.eps);
  };
  const int64_t t_step = (int64_t)gridDim.x * kPmWaves;
  int64_t t = (int64_t)blockIdx.x * kPmWaves + wave;
  bool valid_n = false; int64_t row_n = 0; float pn_n = 0.f;
  if (t < n_tiles) pn_n = coord(t, valid_n, row_n);
  for (; t < n_tiles; t += t_step) {
    const int64_t row0 = t * 32;
    const bool valid = valid_n;
    const int64_t row = row_n;
    const float pn = pn_n;
    if (t + t_step < n_tiles) pn_n = coord(t + t_step, valid_n, row_n);

    if (TRAIN && valid && h == 0) p.pn[row] = pn;        // all the weight gradient needs to regenerate the features

    // ---- GEMM 1: z1[m][row] = sum_k W0[m][k] F[row][k], F generated as B fragments ---------------------------------
    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < kHalf / 16; ++s) {
      const int k0 = 16 * s + 8 * h;
      const f32x4 wa = *reinterpret_cast<const f32x4*>(freqs + k0);
      const f32x4 wb = *reinterpret_cast<const f32x4*>(freqs + k0 + 4);
      const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
      float cs[8], sn[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) sincos_hw(pn * w[j], &sn[j], &cs[j]);
      const u32x4 bc = pack8<T>(cs), bs = pack8<T>(sn);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const unsigned char* wr = lw0 + (ct * 32 + r) * kW0Stride + k0 * 2;
        acc[ct] = Mfma<T>::run(*reinterpret_cast<const u32x4*>(wr), bc, acc[ct]);
        acc[ct] = Mfma<T>::run(*reinterpret_cast<const u32x4*>(wr + kHalf * 2), bs, acc[ct]);
      }
    }

    // ---- z1 = acc + b0 (lane: data row r, columns ct*32 + 8g + 4h + {0..3}), h1 = SiLU(z1), kept as 16-bit pairs (the
    //      form the second GEMM's B operand and the optional store want) ---------------------------------------------------
    uint32_t h1p[2][8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = ct * 32 + 8 * gq + 4 * h;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lb0 + col);
        const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
        float z[4], hq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          z[j] = acc[ct][4 * gq + j] + bq[j];
          hq[j] = z[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z[j]));
        }
        h1p[ct][2 * gq] = Vec8<T>::pack(hq[0], hq[1]);
        h1p[ct][2 * gq + 1] = Vec8<T>::pack(hq[2], hq[3]);
        if (TRAIN) {
          uint2 pk;
          pk.x = Vec8<T>::pack(z[0], z[1]);
          pk.y = Vec8<T>::pack(z[2], z[3]);
          *reinterpret_cast<uint2*>(et + r * kEStride + col * 2) = pk;
        }
      }
    }
    if (TRAIN) {
      __builtin_amdgcn_wave_barrier();                    // the tile is wave-private: one wave's LDS ops stay in order
      store_tile<T>(et, static_cast<T*>(p.z1), row0, n_rows, lane);
      __builtin_amdgcn_wave_barrier();
      if (p.h1) {                                         // (segger_posmlp_bwd recomputes it from z1 instead)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            uint2 pk;
            pk.x = h1p[ct][2 * gq];
            pk.y = h1p[ct][2 * gq + 1];
            *reinterpret_cast<uint2*>(et + r * kEStride + (ct * 32 + 8 * gq + 4 * h) * 2) = pk;
          }
        }
        __builtin_amdgcn_wave_barrier();
        store_tile<T>(et, static_cast<T*>(p.h1), row0, n_rows, lane);
        __builtin_amdgcn_wave_barrier();
      }
    }

    // ---- GEMM 2: pe[m2][row] = sum_m W2[m2][m] h1[m][row]; k-step (ct, gp) covers m = ct*32 + 16gp + {0..15} in the
    //      order the accumulator lanes hold them: lane half h has 4h+{0..3} and 8+4h+{0..3} ----------------------------
    f32x16 acc2[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[ct][i] = 0.f;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const u32x4 b = {h1p[ct][4 * gp], h1p[ct][4 * gp + 1], h1p[ct][4 * gp + 2], h1p[ct][4 * gp + 3]};
        const int base = ct * 32 + 16 * gp + 4 * h;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const unsigned char* wr = lw2 + (c2 * 32 + r) * kW2Stride + base * 2;
          const uint2 a_lo = *reinterpret_cast<const uint2*>(wr);
          const uint2 a_hi = *reinterpret_cast<const uint2*>(wr + 16);
          acc2[c2] = Mfma<T>::run(u32x4{a_lo.x, a_lo.y, a_hi.x, a_hi.y}, b, acc2[c2]);
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lb2 + ct * 32 + 8 * gq + 4 * h);
        acc2[ct][4 * gq] += bv.x; acc2[ct][4 * gq + 1] += bv.y; acc2[ct][4 * gq + 2] += bv.z; acc2[ct][4 * gq + 3] += bv.w;
      }
    }
    auto put_tile = [&](bool act) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int col = ct * 32 + 8 * gq + 4 * h;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = act ? gelu_erf(acc2[ct][4 * gq + j]) : acc2[ct][4 * gq + j];
          uint2 pk;
          pk.x = Vec8<T>::pack(v[0], v[1]);
          pk.y = Vec8<T>::pack(v[2], v[3]);
          *reinterpret_cast<uint2*>(et + r * kEStride + col * 2) = pk;
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    if (GELU && TRAIN) {                                  // the backward needs the pre-activation for gelu'
      put_tile(false);
      store_tile<T>(et, static_cast<T*>(p.pe_pre), row0, n_rows, lane);
      __builtin_amdgcn_wave_barrier();
    }
    put_tile(GELU);
    store_tile<T>(et, pe, row0, n_rows, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_posmlp_supported(int32_t freq_dim, int32_t dim, int32_t dtype) {
  return freq_dim == kFreq && dim == kDim && (dtype == SEGGER_BF16 || dtype == SEGGER_F16);
}

extern "C" int segger_posmlp_fwd(const float* pos, const int64_t* batch, const float* mins, const float* maxs, int64_t n,
                                 float eps, float max_period, const void* w0, const float* b0, const void* w2,
                                 const float* b2, void* pe, void* z1, float* pn, void* h1, void* pe_pre, int32_t gelu,
                                 int32_t dtype, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0, "segger_posmlp_fwd: negative size");
  SEGGER_REQUIRE(dtype == SEGGER_BF16 || dtype == SEGGER_F16, "segger_posmlp_fwd: bf16 / f16 only");
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(pos && mins && maxs && w0 && b0 && w2 && b2 && pe, "segger_posmlp_fwd: NULL pointer");
  SEGGER_REQUIRE(!z1 == !pn && (z1 || !h1),
                 "segger_posmlp_fwd: z1 and pn go together (both for training, none for inference); h1 only with them");
  SEGGER_REQUIRE(aligned16(h1), "segger_posmlp_fwd: h1 must be 16-byte aligned");
  SEGGER_REQUIRE(aligned16(w0) && aligned16(w2) && aligned16(pe) && aligned16(z1) && aligned16(pn),
                 "segger_posmlp_fwd: matrices must be 16-byte aligned");
  SEGGER_REQUIRE(!pe_pre || (gelu && z1), "segger_posmlp_fwd: pe_pre is the training output of the gelu variant");
  SEGGER_REQUIRE(!(gelu && z1) || pe_pre, "segger_posmlp_fwd: training with gelu needs pe_pre");
  SEGGER_REQUIRE(aligned16(pe_pre), "segger_posmlp_fwd: pe_pre must be 16-byte aligned");
  PosMlpParams p{pos, batch, mins, maxs, n, eps, logf(max_period), w0, b0, w2, b2, pe, z1, pn, pe_pre, h1};
  const int64_t n_tiles = (2 * n + 31) / 32;
  int64_t blocks = (n_tiles + kPmWaves - 1) / kPmWaves;
  if (blocks > 512) blocks = 512;                          // persistent: 2 workgroups per CU
  const bool train = z1 != nullptr;
#define GO(T)                                                                                                    \
  do {                                                                                                           \
    if (train && gelu) hipLaunchKernelGGL((posmlp_fwd_kernel<T, true, true>), dim3((unsigned)blocks), dim3(kPmWaves * 64), 0, stream, p);        \
    else if (train) hipLaunchKernelGGL((posmlp_fwd_kernel<T, true, false>), dim3((unsigned)blocks), dim3(kPmWaves * 64), 0, stream, p);          \
    else if (gelu) hipLaunchKernelGGL((posmlp_fwd_kernel<T, false, true>), dim3((unsigned)blocks), dim3(kPmWaves * 64), 0, stream, p);           \
    else hipLaunchKernelGGL((posmlp_fwd_kernel<T, false, false>), dim3((unsigned)blocks), dim3(kPmWaves * 64), 0, stream, p);                    \
  } while (0)
  if (dtype == SEGGER_BF16) GO(bf16_t); else GO(f16_t);
#undef GO
  SEGGER_LAUNCH_CHECK("posmlp_fwd_kernel");
  return SEGGER_OK;
}
