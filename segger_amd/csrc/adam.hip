// The optimizer step of the hot path (lightning_model.py:300-303: torch.optim.Adam, default betas / eps, no weight
// decay, no amsgrad) for ALL parameter tensors in one launch, on the optimizer's own state tensors.
//
// torch's fused multi-tensor Adam needs 3 launches for the ~60 small tensors of the encoder (one _foreach_add on the
// step counters + 2 chunked multi_tensor_apply launches: 4.8 + 24 + 14.5 us for 0.3 M parameters -- 3 % of a captured
// 1M-edge training step).  Here: one tiny launch that advances the step counters, one launch whose workgroups find
// their (tensor, offset) by a search over a prefix table carried in the kernel arguments.
#include "common.h"

namespace segger {
namespace {

constexpr int kAdamMaxTensors = 64;
constexpr int kAdamElemsPerBlock = 2048;          // 256 threads x 2 x float4

struct AdamBatch {
  segger_adam_tensor t[kAdamMaxTensors];
  int32_t first_block[kAdamMaxTensors + 1];       // prefix sums of the tensors' block counts
  int32_t n;
  double lr, beta1, beta2, eps;                   // torch keeps them as Python floats and evaluates 1 - beta, beta^step in double
  int64_t* counter; int64_t counter_inc;          // optional: *counter += counter_inc by the first workgroup (nobody reads it here)
  const double* hyper;                            // optional DEVICE {lr, beta1, beta2, eps}: overrides the four values above
};

__global__ __launch_bounds__(64) void adam_advance_kernel(AdamBatch b) {
  for (int i = threadIdx.x; i < b.n; i += 64) *b.t[i].step += 1.0f;
}

__global__ __launch_bounds__(256) void adam_kernel(AdamBatch b) {
  const int blk = blockIdx.x;
  int lo = 0, hi = b.n;                            // last tensor with first_block <= blk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b.first_block[mid] <= blk) lo = mid; else hi = mid;
  }
  if (blk == 0 && threadIdx.x == 0 && b.counter != nullptr) *b.counter += b.counter_inc;
  const segger_adam_tensor& t = b.t[lo];
  const int64_t base = (int64_t)(blk - b.first_block[lo]) * kAdamElemsPerBlock;
  // torch/optim/adam.py (capturable, fused=False formulas; the fused kernel evaluates the same expressions in fp32):
  //   exp_avg = lerp(exp_avg, grad, 1 - beta1);  exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad^2
  //   param -= (lr / (1 - beta1^step)) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps)
  const double step = (double)*t.step;             // already advanced by adam_advance_kernel
  const double lr = b.hyper ? b.hyper[0] : b.lr, beta1_ = b.hyper ? b.hyper[1] : b.beta1;
  const double beta2_ = b.hyper ? b.hyper[2] : b.beta2, eps_ = b.hyper ? b.hyper[3] : b.eps;
  const float bc1 = (float)(1.0 - pow(beta1_, step));
  const float bc2 = (float)(1.0 - pow(beta2_, step));
  const float step_size = (float)lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  const float w1 = (float)(1.0 - beta1_), w2 = (float)(1.0 - beta2_), beta2 = (float)beta2_, eps = (float)eps_;
  auto one = [&](float& p, float g, float& m, float& v) {
    m = m + (g - m) * w1;
    v = beta2 * v + w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
  };
  const bool vec = (t.numel % 4 == 0) &&
                   (((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15u) == 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int64_t i = base + ((int64_t)r * 256 + threadIdx.x) * 4;
    if (i >= t.numel) break;
    if (vec) {
      f32x4 p = *reinterpret_cast<f32x4*>(t.param + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(t.grad + i);
      f32x4 m = *reinterpret_cast<f32x4*>(t.exp_avg + i);
      f32x4 v = *reinterpret_cast<f32x4*>(t.exp_avg_sq + i);
      float pp[4] = {p.x, p.y, p.z, p.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
      const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) one(pp[k], gg[k], mm[k], vv[k]);
      p = f32x4{pp[0], pp[1], pp[2], pp[3]}; m = f32x4{mm[0], mm[1], mm[2], mm[3]}; v = f32x4{vv[0], vv[1], vv[2], vv[3]};
      *reinterpret_cast<f32x4*>(t.param + i) = p;
      *reinterpret_cast<f32x4*>(t.exp_avg + i) = m;
      *reinterpret_cast<f32x4*>(t.exp_avg_sq + i) = v;
    } else {
      for (int k = 0; k < 4 && i + k < t.numel; ++k) one(t.param[i + k], t.grad[i + k], t.exp_avg[i + k], t.exp_avg_sq[i + k]);
    }
  }
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_adam_step(const segger_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                                double eps, segger_stream_t stream_) {
  return segger_adam_step_ex(tensors, n_tensors, lr, beta1, beta2, eps, 0, nullptr, 0, stream_);
}

static int adam_launch(const segger_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps,
                       const double* hyper, int32_t flags, int64_t* counter, int64_t counter_inc, segger_stream_t stream_);

extern "C" int segger_adam_step_ex(const segger_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                                   double eps, int32_t flags, int64_t* counter, int64_t counter_inc, segger_stream_t stream_) {
  SEGGER_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.,
                 "segger_adam_step: lr / betas / eps out of range");
  return adam_launch(tensors, n_tensors, lr, beta1, beta2, eps, nullptr, flags, counter, counter_inc, stream_);
}

extern "C" int segger_adam_step_dev(const segger_adam_tensor* tensors, int32_t n_tensors, const double* hyper, int32_t flags,
                                    int64_t* counter, int64_t counter_inc, segger_stream_t stream_) {
  SEGGER_REQUIRE(hyper != nullptr && ((uintptr_t)hyper % 8) == 0, "segger_adam_step_dev: hyper is NULL or misaligned");
  return adam_launch(tensors, n_tensors, 0., 0., 0., 0., hyper, flags, counter, counter_inc, stream_);
}

static int adam_launch(const segger_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps,
                       const double* hyper, int32_t flags, int64_t* counter, int64_t counter_inc, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bool advanced = (flags & SEGGER_ADAM_STEPS_ADVANCED) != 0;
  bool counter_done = counter == nullptr;
  SEGGER_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || tensors), "segger_adam_step: NULL tensor table");
  for (int32_t at = 0; at < n_tensors; at += kAdamMaxTensors) {
    AdamBatch b{};
    b.n = n_tensors - at < kAdamMaxTensors ? n_tensors - at : kAdamMaxTensors;
    b.lr = lr; b.beta1 = beta1; b.beta2 = beta2; b.eps = eps; b.hyper = hyper;
    int64_t blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const segger_adam_tensor& t = tensors[at + i];
      SEGGER_REQUIRE(t.numel >= 0 && t.step, "segger_adam_step: tensor %d: bad size or NULL step", at + i);
      SEGGER_REQUIRE(t.numel == 0 || (t.param && t.grad && t.exp_avg && t.exp_avg_sq), "segger_adam_step: tensor %d: NULL pointer", at + i);
      b.t[i] = t;
      b.first_block[i] = (int32_t)blocks;
      blocks += (t.numel + kAdamElemsPerBlock - 1) / kAdamElemsPerBlock;
      SEGGER_REQUIRE(blocks < 0x7fffffffLL, "segger_adam_step: too many elements");
    }
    b.first_block[b.n] = (int32_t)blocks;
    if (!advanced) hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, stream, b);
    if (blocks > 0) {
      if (!counter_done) { b.counter = counter; b.counter_inc = counter_inc; counter_done = true; }
      hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b);
    }
    SEGGER_LAUNCH_CHECK("adam kernels");
  }
  if (!counter_done) return segger_step_advance(counter, counter_inc, nullptr, stream_);     // (nothing to update)
  return SEGGER_OK;
}
