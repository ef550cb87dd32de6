// Generic GATv2 kernels for (heads, channels) combinations without a specialised
// geometry (gatv2_launch.h lists those): any H >= 1, 1 <= C <= 512, any row stride.
// One wave per CSR row; heads are processed one after another; lanes stride the
// channels of the head (element loads, no alignment requirement).  Same arithmetic
// as the specialised kernels (base-2 online softmax, lse, D trick, dropout mask),
// written for coverage rather than speed.  grad_att / grad_bias are accumulated with
// fp32 atomics into buffers the launcher zero-fills.
#include "gatv2_launch.h"

namespace segger {
namespace {

template <typename T> __device__ __forceinline__ float ldg(const T* p);
template <> __device__ __forceinline__ float ldg<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldg<bf16_t>(const bf16_t* p) { return __uint_as_float((uint32_t)p->v << 16); }
template <> __device__ __forceinline__ float ldg<f16_t>(const f16_t* p) { return static_cast<float>(__builtin_bit_cast(_Float16, p->v)); }
template <typename T> __device__ __forceinline__ void stg(T* p, float v);
template <> __device__ __forceinline__ void stg<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stg<bf16_t>(bf16_t* p, float v) { p->v = (uint16_t)(Vec8<bf16_t>::pack(v, 0.f) & 0xffffu); }
template <> __device__ __forceinline__ void stg<f16_t>(f16_t* p, float v) { p->v = (uint16_t)(Vec8<f16_t>::pack(v, 0.f) & 0xffffu); }

constexpr int kMaxCPL = 8;     // channels per lane: C <= 512

struct GenParams {
  GatParams g;
  int H, C;
  float* grad_att;   // bwd dst: atomically accumulated
  float* grad_bias;
};

__device__ __forceinline__ float lrelu(float t, float slope) { return t > 0.f ? t : slope * t; }

template <typename T>
__global__ __launch_bounds__(256) void gen_fwd_kernel(GenParams q) {
  const GatParams& p = q.g;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.n_rows) return;
  const int H = q.H, C = q.C;
  const T* xl = static_cast<const T*>(p.xl);
  const T* xr = static_cast<const T*>(p.xr);
  const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
  const bool dropout = p.drop_thr != 0;
  uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
  if (dropout && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    seed_lo = (uint32_t)mixed; seed_hi = (uint32_t)(mixed >> 32);
  }
  for (int h = 0; h < H; ++h) {
    float xr_c[kMaxCPL], att_c[kMaxCPL], acc[kMaxCPL];
#pragma unroll
    for (int k = 0; k < kMaxCPL; ++k) {
      const int c = lane + 64 * k;
      xr_c[k] = c < C ? ldg(xr + row * p.ld_xr + h * C + c) : 0.f;
      att_c[k] = c < C ? p.att[h * C + c] : 0.f;
      acc[k] = 0.f;
    }
    float m = -INFINITY, s = 0.f;
    for (int64_t e = beg; e < end; ++e) {
      const int64_t i = p.col[e];
      float v[kMaxCPL], pl = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCPL; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < C ? ldg(xl + i * p.ld_xl + h * C + c) : 0.f;
        pl = fmaf(att_c[k], lrelu(v[k] + xr_c[k], p.slope), pl);
      }
      const float e2 = wave_sum(pl) * kLog2e;
      const float mn = fmaxf(m, e2);
      const float sc = fast_exp2(m - mn), pe = fast_exp2(e2 - mn);
      s = s * sc + pe;
      float w = pe;
      const int64_t id = p.eid ? (int64_t)p.eid[e] : e;
      if (dropout) w = dropout_keep((uint32_t)id, H, h, seed_lo, seed_hi, p.drop_thr) ? pe * p.drop_scale : 0.f;
      if (p.alpha && lane == 0) p.alpha[id * H + h] = e2;
#pragma unroll
      for (int k = 0; k < kMaxCPL; ++k) acc[k] = acc[k] * sc + w * v[k];
      m = mn;
    }
    const float lse = m + fast_log2(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
#pragma unroll
    for (int k = 0; k < kMaxCPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        float o = acc[k] * inv + (p.bias ? p.bias[h * C + c] : 0.f);
        if (p.pre && (p.pre != p.out || p.apply_gelu)) stg(static_cast<T*>(p.pre) + row * p.ld_pre + h * C + c, o);
        if (p.apply_gelu) o = gelu_erf(o);
        stg(static_cast<T*>(p.out) + row * p.ld_out + h * C + c, o);
      }
    }
    if (p.lse && lane == 0) p.lse[row * H + h] = lse;
    if (p.alpha && lane == 0) {
      for (int64_t e = beg; e < end; ++e) {
        const int64_t id = p.eid ? (int64_t)p.eid[e] : e;
        float a = fast_exp2(p.alpha[id * H + h] - lse);
        if (dropout) a = dropout_keep((uint32_t)id, H, h, seed_lo, seed_hi, p.drop_thr) ? a * p.drop_scale : 0.f;
        p.alpha[id * H + h] = a;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gen_bwd_dst_kernel(GenParams q) {
  const GatParams& p = q.g;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.n_rows) return;
  const int H = q.H, C = q.C;
  const T* xl = static_cast<const T*>(p.xl);
  const T* xr = static_cast<const T*>(p.xr);
  const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
  const bool dropout = p.drop_thr != 0;
  uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
  if (dropout && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    seed_lo = (uint32_t)mixed; seed_hi = (uint32_t)(mixed >> 32);
  }
  for (int h = 0; h < H; ++h) {
    float xr_c[kMaxCPL], att_c[kMaxCPL], g[kMaxCPL], dxr[kMaxCPL], datt[kMaxCPL];
    float D = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxCPL; ++k) {
      const int c = lane + 64 * k;
      xr_c[k] = att_c[k] = g[k] = dxr[k] = datt[k] = 0.f;
      if (c < C) {
        const int64_t o = h * C + c;
        xr_c[k] = ldg(xr + row * p.ld_xr + o);
        att_c[k] = p.att[o];
        const float gy = ldg(static_cast<const T*>(p.gout) + row * p.ld_go + o);
        const float pr = ldg(static_cast<const T*>(p.pre) + row * p.ld_pre + o);
        g[k] = p.apply_gelu ? gy * gelu_erf_grad(pr) : gy;
        D = fmaf(g[k], pr - (p.bias ? p.bias[o] : 0.f), D);
        stg(static_cast<T*>(p.gpre) + row * p.ld_gp + o, g[k]);
        if (q.grad_bias) atomicAdd(q.grad_bias + o, g[k]);
      }
    }
    D = wave_sum(D);
    if (lane == 0) p.dsum[row * H + h] = D;
    const float lse = p.lse[row * H + h];
    for (int64_t e = beg; e < end; ++e) {
      const int64_t i = p.col[e];
      float v[kMaxCPL], pl = 0.f, da = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCPL; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < C ? ldg(xl + i * p.ld_xl + h * C + c) : 0.f;
        pl = fmaf(att_c[k], lrelu(v[k] + xr_c[k], p.slope), pl);
        da = fmaf(g[k], v[k], da);
      }
      pl = wave_sum(pl); da = wave_sum(da);
      const float a = fast_exp2(pl * kLog2e - lse);
      const int64_t id = p.eid ? (int64_t)p.eid[e] : e;
      if (dropout) da = dropout_keep((uint32_t)id, H, h, seed_lo, seed_hi, p.drop_thr) ? da * p.drop_scale : 0.f;
      const float de = a * (da - D);
#pragma unroll
      for (int k = 0; k < kMaxCPL; ++k) {
        const float t = v[k] + xr_c[k];
        dxr[k] = fmaf(de, t > 0.f ? att_c[k] : att_c[k] * p.slope, dxr[k]);
        datt[k] = fmaf(de, lrelu(t, p.slope), datt[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxCPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        stg(static_cast<T*>(p.gxr) + row * p.ld_gxr + h * C + c, dxr[k]);
        atomicAdd(q.grad_att + h * C + c, datt[k]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gen_bwd_src_kernel(GenParams q) {
  const GatParams& p = q.g;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.n_rows) return;
  const int H = q.H, C = q.C;
  const T* xl = static_cast<const T*>(p.xl);
  const T* xr = static_cast<const T*>(p.xr);
  const T* gp = static_cast<const T*>(p.gpre);
  const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
  const bool dropout = p.drop_thr != 0;
  uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
  if (dropout && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    seed_lo = (uint32_t)mixed; seed_hi = (uint32_t)(mixed >> 32);
  }
  for (int h = 0; h < H; ++h) {
    float v[kMaxCPL], att_c[kMaxCPL], acc[kMaxCPL];
#pragma unroll
    for (int k = 0; k < kMaxCPL; ++k) {
      const int c = lane + 64 * k;
      v[k] = c < C ? ldg(xl + row * p.ld_xl + h * C + c) : 0.f;
      att_c[k] = c < C ? p.att[h * C + c] : 0.f;
      acc[k] = 0.f;
    }
    for (int64_t e = beg; e < end; ++e) {
      const int64_t j = p.col[e];
      float xr_c[kMaxCPL], g[kMaxCPL], pl = 0.f, da = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxCPL; ++k) {
        const int c = lane + 64 * k;
        xr_c[k] = c < C ? ldg(xr + j * p.ld_xr + h * C + c) : 0.f;
        g[k] = c < C ? ldg(gp + j * p.ld_gp + h * C + c) : 0.f;
        pl = fmaf(att_c[k], lrelu(v[k] + xr_c[k], p.slope), pl);
        da = fmaf(g[k], v[k], da);
      }
      pl = wave_sum(pl); da = wave_sum(da);
      const float a = fast_exp2(pl * kLog2e - p.lse[j * H + h]);
      float a_eff = a;
      const int64_t id = p.eid ? (int64_t)p.eid[e] : e;
      if (dropout) {
        const bool keep = dropout_keep((uint32_t)id, H, h, seed_lo, seed_hi, p.drop_thr);
        da = keep ? da * p.drop_scale : 0.f;
        a_eff = keep ? a * p.drop_scale : 0.f;
      }
      const float de = a * (da - p.dsum[j * H + h]);
#pragma unroll
      for (int k = 0; k < kMaxCPL; ++k) {
        const float t = v[k] + xr_c[k];
        acc[k] = fmaf(a_eff, g[k], acc[k]);
        acc[k] = fmaf(de, t > 0.f ? att_c[k] : att_c[k] * p.slope, acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxCPL; ++k) {
      const int c = lane + 64 * k;
      if (c < C) stg(static_cast<T*>(p.gxl) + row * p.ld_gxl + h * C + c, acc[k]);
    }
  }
}

template <typename T>
int launch_generic(int pass, const GenParams& q, hipStream_t stream) {
  const int64_t nb = (q.g.n_rows + 3) / 4;
  if (nb == 0) return SEGGER_OK;
  if (nb > 0x7fffffffLL) { set_error("gatv2 (generic): too many rows"); return SEGGER_EUNSUPPORTED; }
  dim3 grid((unsigned)nb), block(256);
  if (pass == 0) hipLaunchKernelGGL((gen_fwd_kernel<T>), grid, block, 0, stream, q);
  else if (pass == 1) hipLaunchKernelGGL((gen_bwd_dst_kernel<T>), grid, block, 0, stream, q);
  else hipLaunchKernelGGL((gen_bwd_src_kernel<T>), grid, block, 0, stream, q);
  SEGGER_LAUNCH_CHECK("gatv2 generic kernel");
  return SEGGER_OK;
}

}  // namespace

bool gatv2_has_specialised(int heads, int channels) {
#define X(H, LPH) if (heads == H && channels == LPH * 8) return true;
  SEGGER_GEOMETRIES(X)
#undef X
  return false;
}

// pass: 0 fwd, 1 bwd-dst (grad_att / grad_bias zero-filled here, then atomically accumulated), 2 bwd-src
int gatv2_launch_generic(int pass, GatParams& p, int dtype, int heads, int channels, float* grad_att, float* grad_bias,
                         hipStream_t stream) {
  if (channels > 64 * kMaxCPL) {
    set_error("gatv2: channels=%d exceeds the generic kernel's limit of %d", channels, 64 * kMaxCPL);
    return SEGGER_EUNSUPPORTED;
  }
  GenParams q{p, heads, channels, grad_att, grad_bias};
  if (pass == 1) {
    SEGGER_HIP(hipMemsetAsync(grad_att, 0, (size_t)heads * channels * sizeof(float), stream));
    if (grad_bias) SEGGER_HIP(hipMemsetAsync(grad_bias, 0, (size_t)heads * channels * sizeof(float), stream));
  }
  switch (dtype) {
    case SEGGER_F32: return launch_generic<float>(pass, q, stream);
    case SEGGER_BF16: return launch_generic<bf16_t>(pass, q, stream);
    case SEGGER_F16: return launch_generic<f16_t>(pass, q, stream);
    default: set_error("gatv2: unknown dtype %d", dtype); return SEGGER_EINVAL;
  }
}

}  // namespace segger
