// Shared device/host helpers for libsegger_amd (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/segger_amd.h"

namespace segger {

// ---------------------------------------------------------------- errors ---
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define SEGGER_REQUIRE(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      ::segger::set_error(__VA_ARGS__);           \
      return SEGGER_EINVAL;                       \
    }                                             \
  } while (0)

#define SEGGER_HIP(call)                                        \
  do {                                                          \
    hipError_t _e = (call);                                     \
    if (_e != hipSuccess) return ::segger::hip_fail(_e, #call); \
  } while (0)

#define SEGGER_LAUNCH_CHECK(name)                                   \
  do {                                                              \
    hipError_t _e = hipGetLastError();                              \
    if (_e != hipSuccess) return ::segger::hip_fail(_e, name);      \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Compute units of the CURRENT device, cached per device id (a process may drive several GPUs: a persistent grid sized
// from whichever device asked first would be wrong on a heterogeneous node).  0 = unknown.
inline int device_cu_count() {
  static int cache[64] = {};            // written once per device with the same value: a benign race
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cache[dev] > 0) return cache[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
  if (dev < 64 && n > 0) cache[dev] = n;
  return n;
}

// ------------------------------------------------- fp32 projections (csrc/linear_f32.hip) ---
int linear_f32_launch(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t n_rows,
                      int k_in, int m_out, hipStream_t stream, const float* rowbias = nullptr, const int32_t* rowidx = nullptr,
                      int64_t ld_rb = 0, const float* gate = nullptr, int64_t ld_gate = 0, int gate_kind = 0,
                      float* y_act = nullptr, int64_t ld_yact = 0, int act_kind = 0);
size_t wgrad_f32_workspace_bytes(int64_t n_rows, int m_out, int k_in);
// csrc/linear_f32_split.hip: the same weight gradient as six bf16 partial products (same partial-sum layout)
bool wgrad_f32_split_shape_ok(int m, int k);
int64_t wgrad_f32_split_grid(int64_t n_rows, int m, int k);
int wgrad_f32_split_launch(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t n_rows, int m_out, int k_in,
                           float* partial, int64_t* n_slabs, hipStream_t stream);
int wgrad_f32_launch(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, int64_t n_rows, int m_out, int k_in,
                     float* partial, int64_t* n_slabs, hipStream_t stream);

// ------------------------------------------------- deferred partial sums ---
// out0[e] (e < split) / out1[e - split] (e >= split; out1 may be NULL) = sum_s partial[s * width + e], slabs in order.
// Between segger_reductions_defer_begin() and segger_reductions_flush() (csrc/reduce.hip) the kernels that leave
// per-workgroup partial sums (weight gradients, grad_att / grad_bias slabs) queue their final sum here instead of
// launching it; defer_reduce() returns false when nothing is being deferred (or the sum is too long for one pass,
// or the table is full, or the producer runs on another device / stream than the bracket's) and the caller launches its
// own reduction as usual.
struct ReduceSeg {
  const float* partial; int64_t n_slabs; int64_t width; int64_t split; float* out0; float* out1;
  float* scratch;      // >= kReduceGroups * width floats behind the partials (sums longer than one pass go through it)
};
constexpr int kReduceGroups = 32;
bool defer_reduce(const ReduceSeg& seg, hipStream_t stream);

constexpr int kWave = 64;
constexpr int kNumXcd = 8;

// ------------------------------------------------------- element types -----
struct bf16_t { uint16_t v; };
struct f16_t  { uint16_t v; };

// A pair of fp32 channels.  As an ext vector hipcc selects v_pk_add/mul/fma_f32 for its arithmetic; measured on
// gfx950 (tools/valu_probe.hip, >= 2 waves per SIMD) one v_pk_*_f32 costs 6.1 cycles against 2 x 2.3-2.6 for the
// two scalar instructions it replaces, so the kernels can also be built on a plain struct (SEGGER_SCALAR_PAIRS).
#ifdef SEGGER_SCALAR_PAIRS
struct f32x2 {
  float x, y;
};
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2 operator*(f32x2 a, float b) { return f32x2{a.x * b, a.y * b}; }
__device__ __forceinline__ f32x2 operator-(f32x2 a) { return f32x2{-a.x, -a.y}; }
#else
typedef float    f32x2 __attribute__((ext_vector_type(2)));
#endif
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16   b16x2 __attribute__((ext_vector_type(2)));

// Load / store 8 consecutive channels as fp32.  p must be 16-byte aligned.
template <typename T> struct Vec8;

template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&f)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&f)[8]) {
    f32x4 a = {f[0], f[1], f[2], f[3]};
    f32x4 b = {f[4], f[5], f[6], f[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};

template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
    u32x4 r = *reinterpret_cast<const u32x4*>(p);
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void unpack2(uint32_t w, float& a, float& b) {
    a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f v = {a, b};
    b16x2 h = __builtin_convertvector(v, b16x2);
    return __builtin_bit_cast(uint32_t, h);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
    u32x4 r = {pack(f[0], f[1]), pack(f[2], f[3]), pack(f[4], f[5]), pack(f[6], f[7])};
    *reinterpret_cast<u32x4*>(p) = r;
  }
};

template <> struct Vec8<f16_t> {
  static __device__ __forceinline__ void unpack(uint32_t w, float& a, float& b) {
    h16x2 h = __builtin_bit_cast(h16x2, w);
    a = static_cast<float>(h.x); b = static_cast<float>(h.y);
  }
  static __device__ __forceinline__ void unpack2(uint32_t w, float& a, float& b) { unpack(w, a, b); }
  static __device__ __forceinline__ void load(const f16_t* p, float (&f)[8]) {
    u32x4 r = *reinterpret_cast<const u32x4*>(p);
    unpack(r.x, f[0], f[1]); unpack(r.y, f[2], f[3]);
    unpack(r.z, f[4], f[5]); unpack(r.w, f[6], f[7]);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    h16x2 h = {static_cast<_Float16>(a), static_cast<_Float16>(b)};
    return __builtin_bit_cast(uint32_t, h);
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&f)[8]) {
    u32x4 r = {pack(f[0], f[1]), pack(f[2], f[3]), pack(f[4], f[5]), pack(f[6], f[7])};
    *reinterpret_cast<u32x4*>(p) = r;
  }
};

// ------------------------------------------------------- cross-lane --------
// DPP within a 16-lane row (ctrl must be a compile-time constant).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// (bound_ctrl set: every control used here reads a lane of the same, fully active row, so the `old` operand is never
// taken -- with bound_ctrl clear the compiler materialised it as a v_mov_b32 0 in front of every v_mov_b32_dpp:
// 13 of the forward's ~230 instructions per 4-edge batch)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <-> 7-i within 8
constexpr int kDppMirror = 0x140;     // lane i <-> 15-i within 16
constexpr int kDppRowBcast0 = 0x150;  // row_newbcast:n (gfx90a+): lane n of each row -> whole row

// Sum over aligned blocks of N consecutive lanes (N = 1,2,4,8,16), result in every lane.
template <int N>
__device__ __forceinline__ float lane_block_sum(float v) {
  if constexpr (N >= 2)  v += dpp_f<kDppXor1>(v);
  if constexpr (N >= 4)  v += dpp_f<kDppXor2>(v);
  if constexpr (N >= 8)  v += dpp_f<kDppHalfMirror>(v);
  if constexpr (N >= 16) v += dpp_f<kDppMirror>(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = lane_block_sum<16>(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// ------------------------------------------------------- math --------------
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// exact (erf) GELU, torch.nn.functional.gelu(approximate='none')
// Standard normal CDF without erff (ocml's erff is ~45 VALU ops with two branches; GELU is evaluated for every
// output channel of every layer, inside kernels that are VALU-bound):
//   erfc(|x|/sqrt2) = exp2(s * Q(s)),  s = |x|,  Phi(x) = x < 0 ? erfc/2 : 1 - erfc/2.
// Q is the degree-7 minimax fit of log2(erfc(s/sqrt2))/s on [0, 4 sqrt2], weighted by the error it causes in
// erf (fit error 1.6e-8); evaluated in fp32 the GELU built on it is within 4e-7 (absolute) of the exact
// value on [-8, 8] -- torch's own fp32 gelu is within 1.2e-6.  (The fit bounds the ABSOLUTE error; the relative
// error of the vanishing negative tail, |gelu| < 1e-5, reaches 1 %.)
// Shaped for the VALU's issue costs (profiles/r02_valu_probe.txt: v_max / v_min / v_cmp / v_cndmask cost 4+ cycles and a
// select on vcc needs wait states; an fma with |x| as a source modifier costs 2.5): the argument is |x| in every fma (no
// explicit abs), there is NO clamp -- beyond 4 sqrt2 the fit keeps falling (s Q(s) <= -26, monotone: exp2 -> 0; +-inf
// gives Q = -inf, exp2 = 0) -- and no compare / select: with d = Phi(|x|) - 1/2 = 1/2 - erfc/2 in [0, 1/2],
//   gelu(x)  = x Phi(x)      = x/2 + |x| d                  (x sign(x) = |x|)
//   Phi(x)   = 1/2 + copysign(d, x)                           (one v_bfi_b32)
// (Measured: profiles/r06_aggregation_valu_budget.txt.)  The negative tail is
// formed by cancellation (x/2 + |x| d): absolute error <= |x| 3e-8, inside the 4e-7 bound above.
__device__ __forceinline__ float normal_cdf_minus_half_abs(float x) {     // Phi(|x|) - 1/2
  const float s = __builtin_fabsf(x);
  float q = -2.855192741e-06f;
  q = fmaf(q, s, 3.960562235e-05f);
  q = fmaf(q, s, -1.871826931e-04f);
  q = fmaf(q, s, -1.347308812e-04f);
  q = fmaf(q, s, 7.060847394e-03f);
  q = fmaf(q, s, -5.249462857e-02f);
  q = fmaf(q, s, -4.592086259e-01f);
  q = fmaf(q, s, -1.151105166e+00f);
  return fmaf(__builtin_amdgcn_exp2f(q * s), -0.5f, 0.5f);
}
__device__ __forceinline__ float normal_cdf(float x) { return 0.5f + __builtin_copysignf(normal_cdf_minus_half_abs(x), x); }
__device__ __forceinline__ float gelu_erf(float x) { return fmaf(__builtin_fabsf(x), normal_cdf_minus_half_abs(x), 0.5f * x); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float e = __builtin_amdgcn_exp2f((-0.72134752044448170f * x) * x);
  return fmaf(0.3989422804014327f * x, e, normal_cdf(x));
}

__device__ __forceinline__ float silu_grad(float z) {             // d/dz z sigmoid(z)
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
  return sg * (1.0f + z * (1.0f - sg));
}
__device__ __forceinline__ float act_apply(float z, int kind) {   // kind 1 = GELU (erf form), 2 = SiLU
  return kind == 1 ? gelu_erf(z) : z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
// y *= d act / d gate in a GEMM epilogue: kind 1 = GELU (erf form), 2 = SiLU
__device__ __forceinline__ float gate_grad(float g, int kind) { return kind == 1 ? gelu_erf_grad(g) : silu_grad(g); }

// counter-based attention-dropout mask (include/segger_amd.h): one murmur3-style finaliser per
// (edge, head); seed_lo / seed_hi are the halves of splitmix64(seed), mixed on the host
__device__ __forceinline__ bool dropout_keep(uint32_t eid, uint32_t heads, uint32_t h,
                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t thr) {
  uint32_t x = (eid * heads + h) ^ seed_lo;
  x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; x ^= seed_hi;
  return (x >> 8) >= thr;
}
__host__ __device__ static inline uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// XCD-aware block remap: consecutive logical blocks land on the same XCD (blocks
// are dealt round-robin over the 8 XCDs), so neighbouring rows share one L2.
// Launch a grid padded to a multiple of 8; returns -1 for the padding blocks.
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nblocks_padded, int64_t nblocks) {
  const int64_t per = nblocks_padded / kNumXcd;
  const int64_t b = (bid % kNumXcd) * per + bid / kNumXcd;
  return b < nblocks ? b : -1;
}
static inline int64_t pad_to_xcd(int64_t nblocks) { return (nblocks + kNumXcd - 1) / kNumXcd * kNumXcd; }

}  // namespace segger
