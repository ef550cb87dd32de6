// All random draws of one training step in ONE launch (segger_step_draws): the attention-dropout bit planes of up to four
// edge views, the cluster-aware triplets of loss_tx and loss_bd and the negative boundaries of the segmentation loss.
// Eager, these are dropout_bits_many + 2 x triplet_sample + sample_negatives (4 launches of 5-10 us in a captured 1M-edge
// step); none of them depends on anything the step computes, only on its device-side seed word.
#include "draws.h"

namespace segger {
namespace {

constexpr int kMaxSamplers = 2, kMaxAdvance = 64;
struct StepDraws {
  BitsParams bits[kMaxBitsJobs];
  SampleParams samp[kMaxSamplers];
  NegParams neg;
  int first[kMaxBitsJobs + kMaxSamplers + 2];      // first workgroup of each job (prefix sums; last entry = the grid)
  float* advance[kMaxAdvance]; int n_advance;      // device floats incremented by one (the last workgroup)
};

__global__ __launch_bounds__(256) void step_draws_kernel(StepDraws j) {
  const int b = blockIdx.x;
#pragma unroll
  for (int i = 0; i < kMaxBitsJobs; ++i)
    if (b < j.first[i + 1]) { dropout_bits_body(j.bits[i], b - j.first[i]); return; }
#pragma unroll
  for (int i = 0; i < kMaxSamplers; ++i)
    if (b < j.first[kMaxBitsJobs + i + 1]) { triplet_sample_body(j.samp[i], b - j.first[kMaxBitsJobs + i]); return; }
  if (b < j.first[kMaxBitsJobs + kMaxSamplers + 1]) { sample_negatives_body(j.neg, b - j.first[kMaxBitsJobs + kMaxSamplers]); return; }
  if ((int)threadIdx.x < j.n_advance) *j.advance[threadIdx.x] += 1.0f;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_step_draws(const segger_step_draws_args* a, segger_stream_t stream) {
  SEGGER_REQUIRE(a != nullptr, "segger_step_draws: args is NULL");
  SEGGER_REQUIRE(a->n_bits >= 0 && a->n_bits <= kMaxBitsJobs && (a->n_bits == 0 || a->bits), "segger_step_draws: 0..4 bit-plane jobs");
  SEGGER_REQUIRE(a->n_samplers >= 0 && a->n_samplers <= kMaxSamplers, "segger_step_draws: 0..2 samplers");
  SEGGER_REQUIRE(a->n_bits == 0 || (a->heads > 0 && a->heads <= 8 && a->dropout_p >= 0.f && a->dropout_p < 1.f),
                 "segger_step_draws: heads must be in 1..8 and dropout_p in [0, 1)");
  StepDraws j{};
  int64_t at = 0;
  for (int i = 0; i < kMaxBitsJobs; ++i) {
    j.first[i] = (int)at;
    if (i < a->n_bits) {
      const int rc = fill_bits_params(a->bits[i], a->heads, a->dropout_p, a->seed_dev, i, &j.bits[i]);
      if (rc != SEGGER_OK) return rc;
      at += (a->bits[i].n_edges + 1023) / 1024;
    }
  }
  for (int i = 0; i < kMaxSamplers; ++i) {
    j.first[kMaxBitsJobs + i] = (int)at;
    if (i < a->n_samplers) {
      const segger_sample_job& s = a->samplers[i];
      SEGGER_REQUIRE(s.n >= 0 && s.n_clusters > 0, "segger_step_draws: sampler %d: bad sizes", i);
      SEGGER_REQUIRE(s.n == 0 || (s.lab && s.cdf_pos && s.cdf_neg && s.counts && s.offsets && s.members && s.pos && s.neg),
                     "segger_step_draws: sampler %d: NULL pointer", i);
      SEGGER_REQUIRE(!s.dists == !s.d_pos && !s.d_pos == !s.d_neg, "segger_step_draws: sampler %d: dists, d_pos and d_neg go together", i);
      const uint64_t mixed = splitmix64(s.seed);
      j.samp[i] = SampleParams{s.lab, s.n, s.n_clusters, s.cdf_pos, s.cdf_neg, s.counts, s.offsets, s.members, nullptr,
                               (uint32_t)mixed, (uint32_t)(mixed >> 32), s.seed, a->seed_dev, s.dists, s.pos, s.neg, s.d_pos, s.d_neg};
      at += (s.n + 255) / 256;
    }
  }
  j.first[kMaxBitsJobs + kMaxSamplers] = (int)at;
  SEGGER_REQUIRE(a->neg_n >= 0 && a->neg_n_b >= 0, "segger_step_draws: negative size");
  SEGGER_REQUIRE(a->neg_n == 0 || (a->neg_pos && a->neg_out), "segger_step_draws: NULL pointer (negatives)");
  j.neg = NegParams{a->neg_pos, a->neg_n, a->neg_n_b, a->neg_n_b_dev, a->neg_seed, a->seed_dev, a->neg_out};
  at += (a->neg_n + 255) / 256;
  j.first[kMaxBitsJobs + kMaxSamplers + 1] = (int)at;
  SEGGER_REQUIRE(a->n_advance >= 0 && a->n_advance <= kMaxAdvance && (a->n_advance == 0 || a->advance),
                 "segger_step_draws: 0..64 counters to advance");
  j.n_advance = a->n_advance;
  for (int i = 0; i < a->n_advance; ++i) {
    SEGGER_REQUIRE(a->advance[i] != nullptr, "segger_step_draws: advance[%d] is NULL", i);
    j.advance[i] = a->advance[i];
  }
  if (a->n_advance > 0) at += 1;
  SEGGER_REQUIRE(at < 0x7fffffffLL, "segger_step_draws: too many workgroups");
  if (at == 0) return SEGGER_OK;
  hipLaunchKernelGGL(step_draws_kernel, dim3((unsigned)at), dim3(256), 0, (hipStream_t)stream, j);
  SEGGER_LAUNCH_CHECK("step_draws_kernel");
  return SEGGER_OK;
}
