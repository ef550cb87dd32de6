// The per-step random draws of a training step as device functions (their own kernels in gatv2.hip / heads.hip, and one
// merged launch in step_draws.hip): attention-dropout bit planes, cluster-aware triplet sampling, negative boundaries.
// Every body takes the index of its workgroup inside its own job.
#pragma once
#include "common.h"

namespace segger {

// bits[l][slot] = OR_h keep(eid[slot], h; seed_l) << h for all layers; a thread owns FOUR consecutive slots: one 16-byte
// load of their edge ids and one 4-byte store per plane (byte stores, one slot per thread, ran at 2.3 TB/s)
struct BitsParams { const int32_t* eid; int64_t n_edges; int64_t plane_stride; int heads; uint32_t thr; int n_seeds;
                    int eid_aligned; uint64_t seeds[16]; const uint64_t* seed_dev; uint8_t* bits; };
constexpr int kMaxBitsJobs = 4;
struct BitsJobs { BitsParams job[kMaxBitsJobs]; };       // blockIdx.y picks the job (the edge views of one step)

__device__ __forceinline__ void dropout_bits_body(const BitsParams& p, const int64_t bid) {
  const int64_t s0 = (bid * 256 + threadIdx.x) * 4;
  if (s0 >= p.n_edges) return;
  const bool full = s0 + 3 < p.n_edges;
  uint32_t e[4] = {0u, 0u, 0u, 0u};
  if (full && p.eid_aligned) {                          // (a view sliced out of a slide-level sort may start anywhere)
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.eid + s0);
    e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
  } else {
    for (int k = 0; k < 4 && s0 + k < p.n_edges; ++k) e[k] = (uint32_t)p.eid[s0 + k];
  }
  const uint64_t dev = p.seed_dev ? *p.seed_dev : 0ull;
  for (int l = 0; l < p.n_seeds; ++l) {
    const uint64_t mixed = splitmix64(p.seeds[l] + dev);
    const uint32_t lo = (uint32_t)mixed, hi = (uint32_t)(mixed >> 32);
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t b = 0;
      for (int h = 0; h < p.heads; ++h) b |= (uint32_t)dropout_keep(e[k], p.heads, h, lo, hi, p.thr) << h;
      word |= b << (8 * k);
    }
    uint8_t* plane = p.bits + (int64_t)l * p.plane_stride + s0;
    if (full) *reinterpret_cast<uint32_t*>(plane) = word;          // (plane_stride % 4 == 0: checked by the host)
    else for (int k = 0; k < 4 && s0 + k < p.n_edges; ++k) plane[k] = (uint8_t)(word >> (8 * k));
  }
}

// ---- cluster-aware triplet sampling (loss_tx / loss_bd), one thread per node -----------------------------------
struct SampleParams {
  const int64_t* lab; int64_t n; int n_clusters;
  const float* cdf_pos; const float* cdf_neg;
  const int64_t* counts; const int64_t* offsets; const int64_t* members;
  const float* uniforms; uint32_t seed_lo, seed_hi; uint64_t seed_raw; const uint64_t* seed_dev;
  const float* dists; int64_t* pos; int64_t* neg; float* d_pos; float* d_neg;
};

__device__ __forceinline__ float uniform01(uint32_t node, uint32_t draw, uint32_t lo, uint32_t hi) {
  uint32_t x = (node * 4u + draw) ^ lo;
  x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; x ^= hi;
  x *= 0x9e3779b1u; x ^= x >> 15;
  return (float)(x >> 8) * (1.0f / 16777216.0f);          // 24 bits: [0, 1)
}

__device__ __forceinline__ void triplet_sample_body(const SampleParams& p, const int64_t bid) {
  const int64_t i = bid * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int K = p.n_clusters;
  const int64_t r = p.lab[i];
  if (r < 0 || r >= K) {                                   // masked-out node: no triplet
    p.pos[i] = -1; p.neg[i] = -1;
    if (p.d_pos) { p.d_pos[i] = 0.f; p.d_neg[i] = 0.f; }
    return;
  }
  uint32_t lo = p.seed_lo, hi = p.seed_hi;
  if (!p.uniforms && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    lo = (uint32_t)mixed; hi = (uint32_t)(mixed >> 32);
  }
  float u[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) u[d] = p.uniforms ? p.uniforms[(int64_t)d * p.n + i] : uniform01((uint32_t)i, d, lo, hi);
  int64_t pick[2];
  int cl[2];
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const float* row = (side == 0 ? p.cdf_pos : p.cdf_neg) + r * K;
    int c = 0;
    while (c < K - 1 && row[c] < u[2 * side]) ++c;         // first column with cdf >= u (searchsorted, left)
    const int64_t cnt = p.counts[c];
    int64_t w = (int64_t)floorf(u[2 * side + 1] * (float)cnt);
    int64_t slot = p.offsets[c] + w;
    if (slot >= p.n) slot = p.n - 1;
    if (slot < 0) slot = 0;
    pick[side] = p.members[slot];
    cl[side] = c;
  }
  p.pos[i] = pick[0]; p.neg[i] = pick[1];
  if (p.d_pos) { p.d_pos[i] = p.dists[r * K + cl[0]]; p.d_neg[i] = p.dists[r * K + cl[1]]; }
}

// ---- the segmentation loss's negative boundary: (pos + randint(1, n_b)) % n_b, -1 passes through --------------------
struct NegParams { const int64_t* pos; int64_t n; int64_t n_b; const int64_t* n_b_dev; uint64_t seed_raw;
                   const uint64_t* seed_dev; int64_t* neg; };

__device__ __forceinline__ void sample_negatives_body(const NegParams& p, const int64_t bid) {
  const int64_t e = bid * 256 + threadIdx.x;
  if (e >= p.n) return;
  const int64_t ip = p.pos[e];
  if (ip < 0) { p.neg[e] = -1; return; }
  int64_t n_b = p.n_b;
  if (p.n_b_dev) n_b = *p.n_b_dev;
  if (n_b <= 1) { p.neg[e] = 0; return; }
  const uint64_t mixed = splitmix64(p.seed_raw + (p.seed_dev ? *p.seed_dev : 0ull));
  const float u = uniform01((uint32_t)e, (uint32_t)(e >> 30), (uint32_t)mixed, (uint32_t)(mixed >> 32));
  int64_t shift = 1 + (int64_t)floorf(u * (float)(n_b - 1));
  if (shift > n_b - 1) shift = n_b - 1;                    // u * (n_b - 1) may round up to n_b - 1
  p.neg[e] = (ip + shift) % n_b;
}

// host-side fills shared by the single-purpose entry points and segger_step_draws (validation included)
int fill_bits_params(const struct segger_bits_job& jb, int heads, float dropout_p, const uint64_t* seed_dev, int index,
                     BitsParams* out);

}  // namespace segger
