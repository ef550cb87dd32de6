// The three losses of LitISTEncoder.get_losses (lightning_model.py:151-213) after the sampling, one launch each way.
//
// Round 3 ran the loss head as 6 launches forward (three loss kernels, the combination, two zero fills) and 5 backward,
// and loss_tx's backward -- three rows of packed float atomics per triplet into a [n_tx, C] matrix -- sat on the chip's
// atomic ceiling (0.33 of the HBM roofline at C2, the furthest-below kernel of the step).  Here:
//
//   forward   one grid, three block ranges: loss_tx triplets | loss_bd nodes | loss_sg triplets.  Every block leaves a
//             partial sum; a one-workgroup launch behind it adds them in a fixed order, forms the weighted
//             total (segger_loss_combine_fwd's arithmetic) and -- the incoming gradient of a training step being known --
//             the three backward scale factors.  On the way every block zero-fills its slice of the boundary gradient,
//             and the loss_tx range records, per ACTIVE triplet t = (t, p, n), 1/d(t,p) and 1/d(t,n) and threads t into
//             the contribution chains of rows p and n (a counter and one atomicExch per contribution: head[row] ->
//             next[...]).  A chain holds at most kChainCap entries: a row that receives more (a cluster with a handful of
//             members in the batch that many anchors draw from -- the sampler picks clusters by similarity, not by size)
//             becomes HOT: the contribution that finds the chain full is flagged, and the one that fills it allocates an
//             fp32 accumulator row for the hot row.
//   backward  one grid, three block ranges: transcript ROWS | loss_bd nodes | boundaries.  A transcript row r gathers
//             everything that lands on it -- its own anchor term of loss_tx, the positive / negative terms of the
//             triplets chained to it, the anchor term of its segmentation triplet (a transcript anchors at most one:
//             sg_of_tx) -- in registers and STORES the row once, already pushed through the backward of the row
//             normalisation when the un-normalised embeddings are given.  No float atomic touches the transcript matrix,
//             no zero fill, no second matrix; a row is summed in fp32 and rounded once (only the ORDER of a chain -- the
//             order the forward's exchanges happened in -- varies from run to run).  Hot rows: the flagged contributions
//             are added by their OWN triplet's row group into the hot row's fp32 accumulator with atomics (parallel,
//             however skewed the draws), each contributor counts the row's pending counter down and the last one
//             finishes the row (normalisation backward + store).  The chain walk is bounded by kChainCap whatever the
//             state buffers hold: no input can make the kernel spin.  The boundary side
//             (10^2 - 10^4 rows) keeps its fp32 atomics: the boundary's own positives are summed in registers and added
//             once per boundary, sampled negatives and the metric loss add as before.
//
// Row layout: 16 lanes per row, CPL = C / 16 consecutive channels per lane (C in {32, 64, 128}).
#include "common.h"

namespace segger {
namespace {

constexpr int kItemsPerBlock = 64;                 // rows / triplets per 256-thread block: 4 waves x 4 groups x 4 rounds
constexpr int kChainCap = 64;                      // entries a row's chain holds; beyond that the row is "hot"
constexpr int32_t kOverflow = -2;                  // next[code]: the contribution is not chained (hot row)

// ---- CPL consecutive channels as fp32 ---------------------------------------------------------------------------
template <typename T, int CPL> struct Row;
template <int CPL> struct Row<float, CPL> {
  static __device__ __forceinline__ void load(const float* p, float (&f)[CPL]) {
    if constexpr (CPL == 2) { const float2 v = *reinterpret_cast<const float2*>(p); f[0] = v.x; f[1] = v.y; }
    else {
#pragma unroll
      for (int k = 0; k < CPL; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + k);
        f[k] = v.x; f[k + 1] = v.y; f[k + 2] = v.z; f[k + 3] = v.w;
      }
    }
  }
  static __device__ __forceinline__ void store(float* p, const float (&f)[CPL]) {
    if constexpr (CPL == 2) *reinterpret_cast<float2*>(p) = float2{f[0], f[1]};
    else {
#pragma unroll
      for (int k = 0; k < CPL; k += 4) *reinterpret_cast<f32x4*>(p + k) = f32x4{f[k], f[k + 1], f[k + 2], f[k + 3]};
    }
  }
};
template <typename T, int CPL> struct Row16 {      // bf16_t / f16_t
  static __device__ __forceinline__ void load(const T* p, float (&f)[CPL]) {
    if constexpr (CPL == 2) {
      Vec8<T>::unpack2(*reinterpret_cast<const uint32_t*>(p), f[0], f[1]);
    } else if constexpr (CPL == 4) {
      const uint2 v = *reinterpret_cast<const uint2*>(p);
      Vec8<T>::unpack2(v.x, f[0], f[1]); Vec8<T>::unpack2(v.y, f[2], f[3]);
    } else {
      Vec8<T>::load(p, f);
    }
  }
  static __device__ __forceinline__ void store(T* p, const float (&f)[CPL]) {
    if constexpr (CPL == 2) {
      *reinterpret_cast<uint32_t*>(p) = Vec8<T>::pack(f[0], f[1]);
    } else if constexpr (CPL == 4) {
      *reinterpret_cast<uint2*>(p) = uint2{Vec8<T>::pack(f[0], f[1]), Vec8<T>::pack(f[2], f[3])};
    } else {
      Vec8<T>::store(p, f);
    }
  }
};
template <int CPL> struct Row<bf16_t, CPL> : Row16<bf16_t, CPL> {};
template <int CPL> struct Row<f16_t, CPL> : Row16<f16_t, CPL> {};

template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
}
template <> __device__ __forceinline__ void ld2<bf16_t>(const bf16_t* p, float& a, float& b) {
  Vec8<bf16_t>::unpack2(*reinterpret_cast<const uint32_t*>(p), a, b);
}
template <> __device__ __forceinline__ void ld2<f16_t>(const f16_t* p, float& a, float& b) {
  Vec8<f16_t>::unpack2(*reinterpret_cast<const uint32_t*>(p), a, b);
}

__device__ __forceinline__ float softplus_(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + __expf(-x)); }

struct LossHeadParams {
  const void* ztx; int64_t ld_ztx; int64_t n_tx;
  const void* zbd; int64_t ld_zbd; int64_t n_bd;
  const int64_t* tx_pos; const int64_t* tx_neg; float tx_margin, tx_eps;
  const int64_t* bd_pos; const int64_t* bd_neg; const float* bd_dpos; const float* bd_dneg; const float* bd_w; float bd_eps;
  const int64_t* sg_src; const int64_t* sg_pos; const int64_t* sg_neg; int64_t n_sg; float sg_margin, sg_eps; int sg_kind;
  const int64_t* sg_indptr; const int32_t* sg_eid; const int32_t* sg_of_tx;
  const float* a; const float* b; const float* gout; float* out; float* graw;
  float* tx_w; int32_t* state; int32_t* next;     // state: [n_tx][2] = (chain head code + 1, contributions), then the hot-row counter
  int32_t* hot_id; float* hot_acc; int64_t max_hot;
  const void* ytx; int64_t ld_ytx; float norm_eps;
  void* gtx; int64_t ld_gtx; float* gbd;
  float* partial; int32_t* ticket;
  int nb_tx, nb_bd, nb_sg;
  int nb_sg_fin;             // loss_sg blocks of the FORWARD launch (the backward's nb_sg is one block per boundary)
  int defer;                 // the forward left the finishing to the backward launch (segger_loss_head_args.reserved_ & 1)
};

// d loss / d (raw mean i): what segger_loss_combine_bwd computes; with `defer` from the hinted gradient directly (the
// finishing launch that would have written graw did not run)
__device__ __forceinline__ float graw_of(const LossHeadParams& p, int i) {
  return p.defer ? (p.gout[3] * p.b[i] + p.gout[i]) * p.a[i] : p.graw[i];
}

// ---------------------------------------------------------------------------------------------------- forward ----
template <typename T, int CPL>
__global__ __launch_bounds__(256) void loss_head_fwd_kernel(LossHeadParams p) {
  __shared__ float wsum[4];
  constexpr int C = 16 * CPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15, c0 = gl * CPL;
  const int blk = blockIdx.x;
  const T* ztx = static_cast<const T*>(p.ztx);
  const T* zbd = static_cast<const T*>(p.zbd);
  float acc = 0.f;
  if (blk < p.nb_tx) {
    // ---- loss_tx: TripletMarginLoss over (t, tx_pos[t], tx_neg[t]), rows of z_tx (triplet_loss.py:128-160)
    const int64_t base = (int64_t)blk * kItemsPerBlock;
#pragma unroll 1
    for (int i = wave * 4 + grp; i < kItemsPerBlock; i += 16) {
      const int64_t t = base + i;
      bool ok = t < p.n_tx;                                      // group-uniform
      int64_t ip = ok ? p.tx_pos[t] : 0, in = ok ? p.tx_neg[t] : 0;
      if ((uint64_t)ip >= (uint64_t)p.n_tx || (uint64_t)in >= (uint64_t)p.n_tx) { ok = false; ip = in = 0; }
      const int64_t ia = ok ? t : 0;
      float a[CPL], pp[CPL], nn[CPL];
      Row<T, CPL>::load(ztx + ia * p.ld_ztx + c0, a);
      Row<T, CPL>::load(ztx + ip * p.ld_ztx + c0, pp);
      Row<T, CPL>::load(ztx + in * p.ld_ztx + c0, nn);
      float sp = 0.f, sn = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const float dp = a[k] - pp[k] + p.tx_eps, dn = a[k] - nn[k] + p.tx_eps;
        sp = fmaf(dp, dp, sp); sn = fmaf(dn, dn, sn);
      }
      sp = lane_block_sum<16>(sp); sn = lane_block_sum<16>(sn);
      const float dap = sqrtf(sp), dan = sqrtf(sn);
      const float l = dap - dan + p.tx_margin;
      if (ok && gl == 0) acc += fmaxf(l, 0.f);
      if (p.tx_w != nullptr && t < p.n_tx) {
        const bool active = ok && l > 0.f;
        const float wp = active && dap > 0.f ? 1.0f / dap : 0.f;
        const float wn = active && dan > 0.f ? 1.0f / dan : 0.f;
        if (gl < 2) {                                              // lane 0: the positive's row, lane 1: the negative's
          const float wk = gl == 0 ? wp : wn;
          const int64_t tgt = gl == 0 ? ip : in;
          const int32_t code = (int32_t)(2 * t + gl);
          p.tx_w[code] = wk;
          if (wk != 0.f) {
            const int32_t slot = atomicAdd(p.state + 2 * tgt + 1, 1);
            if (slot < kChainCap) {
              p.next[code] = atomicExch(p.state + 2 * tgt, code + 1);        // (codes are stored + 1: 0 ends a chain)
            } else {
              p.next[code] = kOverflow;
              if (slot == kChainCap) {                                         // this one makes the row hot: its accumulator
                const int32_t hid = atomicAdd(p.state + 2 * p.n_tx, 1);
                if ((int64_t)hid < p.max_hot) {
                  float* acc_row = p.hot_acc + (int64_t)hid * C;
                  for (int k = 0; k < C; ++k) acc_row[k] = 0.f;
                  p.hot_id[p.n_tx + hid] = 0;                                  // arrivals at the row in the backward
                  __threadfence();
                }
                p.hot_id[tgt] = hid;
              }
            }
          }
        }
      }
    }
  } else if (blk < p.nb_tx + p.nb_bd) {
    // ---- loss_bd: MetricLoss over the boundaries (triplet_loss.py:163-204)
    const int64_t base = (int64_t)(blk - p.nb_tx) * kItemsPerBlock;
#pragma unroll 1
    for (int q = wave * 4 + grp; q < kItemsPerBlock; q += 16) {
      const int64_t i = base + q;
      bool ok = i < p.n_bd;
      int64_t ip = ok ? p.bd_pos[i] : 0, in = ok ? p.bd_neg[i] : 0;
      const float w = ok ? p.bd_w[i] : 0.f;
      if ((uint64_t)ip >= (uint64_t)p.n_bd || (uint64_t)in >= (uint64_t)p.n_bd || w == 0.f) { ok = false; ip = in = 0; }
      const int64_t ii = ok ? i : 0;
      float x[CPL], yp[CPL], yn[CPL];
      Row<T, CPL>::load(zbd + ii * p.ld_zbd + c0, x);
      Row<T, CPL>::load(zbd + ip * p.ld_zbd + c0, yp);
      Row<T, CPL>::load(zbd + in * p.ld_zbd + c0, yn);
      float sxx = 0.f, spp = 0.f, snn = 0.f, sxp = 0.f, sxn = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        sxx = fmaf(x[k], x[k], sxx); spp = fmaf(yp[k], yp[k], spp); snn = fmaf(yn[k], yn[k], snn);
        sxp = fmaf(x[k], yp[k], sxp); sxn = fmaf(x[k], yn[k], sxn);
      }
      sxx = lane_block_sum<16>(sxx); spp = lane_block_sum<16>(spp); snn = lane_block_sum<16>(snn);
      sxp = lane_block_sum<16>(sxp); sxn = lane_block_sum<16>(sxn);
      const float cx = fmaxf(sqrtf(sxx), p.bd_eps), cp = fmaxf(sqrtf(spp), p.bd_eps), cn = fmaxf(sqrtf(snn), p.bd_eps);
      const float rp = sxp / (cx * cp) - (1.f - (ok ? p.bd_dpos[i] : 0.f));
      const float rn = sxn / (cx * cn) - (1.f - (ok ? p.bd_dneg[i] : 0.f));
      if (ok && gl == 0) acc += w * (rp * rp + rn * rn);
    }
  } else {
    // ---- loss_sg: triplet margin (lightning_model.py:178-187) or BCE on dot-product logits (:190-207)
    const int64_t base = (int64_t)(blk - p.nb_tx - p.nb_bd) * kItemsPerBlock;
#pragma unroll 1
    for (int q = wave * 4 + grp; q < kItemsPerBlock; q += 16) {
      const int64_t e = base + q;
      bool ok = e < p.n_sg;
      int64_t ia = ok ? p.sg_src[e] : 0, ip = ok ? p.sg_pos[e] : 0, in = ok ? p.sg_neg[e] : 0;
      if ((uint64_t)ia >= (uint64_t)p.n_tx || (uint64_t)ip >= (uint64_t)p.n_bd || (uint64_t)in >= (uint64_t)p.n_bd) {
        ok = false; ia = ip = in = 0;
      }
      float a[CPL], pp[CPL], nn[CPL];
      Row<T, CPL>::load(ztx + ia * p.ld_ztx + c0, a);
      Row<T, CPL>::load(zbd + ip * p.ld_zbd + c0, pp);
      Row<T, CPL>::load(zbd + in * p.ld_zbd + c0, nn);
      float s0 = 0.f, s1 = 0.f;
      if (p.sg_kind == SEGGER_LOSS_BCE) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) { s0 = fmaf(a[k], pp[k], s0); s1 = fmaf(a[k], nn[k], s1); }
        s0 = lane_block_sum<16>(s0); s1 = lane_block_sum<16>(s1);
        if (ok && gl == 0) acc += softplus_(-s0) + softplus_(s1);
      } else {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const float dp = a[k] - pp[k] + p.sg_eps, dn = a[k] - nn[k] + p.sg_eps;
          s0 = fmaf(dp, dp, s0); s1 = fmaf(dn, dn, s1);
        }
        s0 = lane_block_sum<16>(s0); s1 = lane_block_sum<16>(s1);
        if (ok && gl == 0) acc += fmaxf(sqrtf(s0) - sqrtf(s1) + p.sg_margin, 0.f);
      }
    }
  }
  // every block clears its slice of the boundary gradient the backward accumulates into
  if (p.gbd != nullptr) {
    const int64_t total = p.n_bd * C;                             // (C % 4 == 0)
    int64_t per = (total + gridDim.x - 1) / gridDim.x;
    per = (per + 3) & ~(int64_t)3;
    const int64_t beg = (int64_t)blk * per, end = beg + per < total ? beg + per : total;
    for (int64_t i = beg + (int64_t)threadIdx.x * 4; i < end; i += 1024)
      *reinterpret_cast<f32x4*>(p.gbd + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  acc = wave_sum(acc);
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[blk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// The three means in a fixed order, the weighted total and -- the incoming gradient of a training step being known -- the
// three backward scale factors: one workgroup, launched right behind the forward kernel.  (Round 4 first let the LAST block
// of the forward do this, found by a ticket counter behind a __threadfence(): on this part a device-scope release writes
// back the whole L2 of the XCD, once per block -- 0.76 of the forward's 0.90 ms at C2 (22k blocks), 30 of 44 us at 44k rows;
// tools/bench_loss_head.py, profiles/r04_loss_head_modes_c2.txt.)
__device__ __forceinline__ void loss_head_finish_body(const LossHeadParams& p) {
  __shared__ float wsum[4];
  __shared__ float raw[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 3; ++r) {      // (no arrays indexed by r: they would live in scratch memory)
    const int lo = r == 0 ? 0 : (r == 1 ? p.nb_tx : p.nb_tx + p.nb_bd);
    const int hi = r == 0 ? p.nb_tx : (r == 1 ? p.nb_tx + p.nb_bd : p.nb_tx + p.nb_bd + p.nb_sg_fin);
    const float scale = r == 0 ? (p.n_tx > 0 ? 1.0f / (float)p.n_tx : 0.f)
                      : r == 1 ? 1.0f
                               : (p.n_sg > 0 ? (p.sg_kind == SEGGER_LOSS_BCE ? 0.5f : 1.0f) / (float)p.n_sg : 0.f);
    float s = 0.f;
    for (int k = lo + (int)threadIdx.x; k < hi; k += 256) s += p.partial[k];
    s = wave_sum(s);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) raw[r] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float t = raw[i] * p.a[i];
      p.out[i] = t;
      total = fmaf(t, p.b[i], total);
    }
    p.out[3] = total;
    if (p.gout != nullptr && p.graw != nullptr) {
#pragma unroll
      for (int i = 0; i < 3; ++i) p.graw[i] = (p.gout[3] * p.b[i] + p.gout[i]) * p.a[i];
    }
  }
}

__global__ __launch_bounds__(256) void loss_head_finish_kernel(LossHeadParams p) { loss_head_finish_body(p); }

// --------------------------------------------------------------------------------------------------- backward ----
template <typename T, int CPL>
__global__ __launch_bounds__(256) void loss_head_bwd_kernel(LossHeadParams p) {
  constexpr int C = 16 * CPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15, c0 = gl * CPL;
  const int blk = blockIdx.x;
  const T* ztx = static_cast<const T*>(p.ztx);
  const T* zbd = static_cast<const T*>(p.zbd);
  if (p.defer && blk == p.nb_tx + p.nb_bd + p.nb_sg) {      // one extra workgroup: the forward's finishing launch, deferred
    loss_head_finish_body(p);
    return;
  }
  if (blk < p.nb_tx) {
    // ---- one transcript row per 16-lane group: everything that lands on row r, summed in registers, stored once
    const float sc_tx = p.n_tx > 0 ? graw_of(p, 0) / (float)p.n_tx : 0.f;
    const float sc_sg = p.n_sg > 0 ? graw_of(p, 2) * (p.sg_kind == SEGGER_LOSS_BCE ? 0.5f : 1.0f) / (float)p.n_sg : 0.f;
    const int64_t base = (int64_t)blk * kItemsPerBlock;
    // the row's gradient through z = y / max(|y|, eps) when y is given (csrc/frontend.hip l2norm_kernel), then the store
    auto finish_row = [&](int64_t row, float (&g)[CPL]) {
      if (p.ytx != nullptr) {
        float y[CPL];
        Row<T, CPL>::load(static_cast<const T*>(p.ytx) + row * p.ld_ytx + c0, y);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) ss = fmaf(y[k], y[k], ss);
        const float nrm = sqrtf(lane_block_sum<16>(ss));
        const float inv = 1.0f / fmaxf(nrm, p.norm_eps);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) dot = fmaf(y[k] * inv, g[k], dot);
        dot = lane_block_sum<16>(dot);
        if (nrm < p.norm_eps) dot = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) g[k] = (g[k] - y[k] * inv * dot) * inv;
      }
      Row<T, CPL>::store(static_cast<T*>(p.gtx) + row * p.ld_gtx + c0, g);
    };
    // a hot row: add `g` to its fp32 accumulator and count the arrival; the row expects its flagged contributions
    // (count - kChainCap of them; the count is not touched here) + its own group: the last to arrive reads the sum back and
    // finishes the row
    auto hot_add = [&](int64_t row, const float (&g)[CPL]) {
      const int64_t hid = p.hot_id[row];
      if ((uint64_t)hid >= (uint64_t)p.max_hot) return;            // (cannot happen: max_hot bounds the number of hot rows)
      float* acc = p.hot_acc + hid * C + c0;
#pragma unroll
      for (int k = 0; k < CPL; ++k) atomicAdd(acc + k, g[k]);
      __threadfence();
      int32_t old = 0;
      if (gl == 0) old = atomicAdd(p.hot_id + p.n_tx + hid, 1);
      old = __shfl(old, lane & 48, 64);                            // lane 0 of the group
      if (old == p.state[2 * row + 1] - kChainCap) {
        __threadfence();
        float t[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) t[k] = __hip_atomic_load(acc + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        finish_row(row, t);
      }
    };
#pragma unroll 1
    for (int i = wave * 4 + grp; i < kItemsPerBlock; i += 16) {
      const int64_t r = base + i;
      if (r >= p.n_tx) continue;                                   // group-uniform
      float a[CPL], g[CPL];
      Row<T, CPL>::load(ztx + r * p.ld_ztx + c0, a);
#pragma unroll
      for (int k = 0; k < CPL; ++k) g[k] = 0.f;
      // (1) anchor of its own loss_tx triplet; a flagged (unchained) contribution of this triplet goes to its hot row
      // (what takes part is decided by the forward's UNSCALED weights: a hot row waits for as many arrivals as the forward
      //  flagged, also when the loss weight -- sc_tx -- is zero or the product underflows; those arrivals then add zeros)
      const float uwp = p.tx_w[2 * r], uwn = p.tx_w[2 * r + 1];
      const float wp = uwp * sc_tx, wn = uwn * sc_tx;
      if (uwp != 0.f || uwn != 0.f) {
        const int64_t ip = p.tx_pos[r], in = p.tx_neg[r];
        float pp[CPL], nn[CPL], dp[CPL], dn[CPL];
        Row<T, CPL>::load(ztx + ip * p.ld_ztx + c0, pp);
        Row<T, CPL>::load(ztx + in * p.ld_ztx + c0, nn);
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          dp[k] = (a[k] - pp[k] + p.tx_eps) * wp; dn[k] = (a[k] - nn[k] + p.tx_eps) * wn;
          g[k] += dp[k] - dn[k];
        }
        if (uwp != 0.f && p.next[2 * r] == kOverflow) {
#pragma unroll
          for (int k = 0; k < CPL; ++k) dp[k] = -dp[k];
          hot_add(ip, dp);
        }
        if (uwn != 0.f && p.next[2 * r + 1] == kOverflow) hot_add(in, dn);
      }
      // (2) positive / negative of the triplets chained to it (code = 2 t + kind, stored + 1; kind 0: r is t's positive).
      // At most kChainCap entries by construction; the bound also holds against whatever a stale buffer contains.
      const int32_t total = p.state[2 * r + 1];
      int32_t cur = p.state[2 * r];
#pragma unroll 1
      for (int it = 0; it < kChainCap && cur > 0 && (int64_t)cur <= 2 * p.n_tx; ++it) {
        const int32_t code = cur - 1;
        const int64_t t = code >> 1;
        const float w = p.tx_w[code] * sc_tx;
        const int32_t nxt = p.next[code];
        float at[CPL];
        Row<T, CPL>::load(ztx + t * p.ld_ztx + c0, at);
        const float sgn = (code & 1) ? w : -w;
#pragma unroll
        for (int k = 0; k < CPL; ++k) g[k] = fmaf(at[k] - a[k] + p.tx_eps, sgn, g[k]);
        cur = nxt;
      }
      // (3) anchor of its segmentation triplet (a transcript lies in at most one boundary)
      if (p.sg_of_tx != nullptr && sc_sg != 0.f) {
        const int64_t e = p.sg_of_tx[r];
        if (e >= 0 && e < p.n_sg) {
          const int64_t j = p.sg_pos[e], jn = p.sg_neg[e];
          if (p.sg_src[e] == r && (uint64_t)j < (uint64_t)p.n_bd && (uint64_t)jn < (uint64_t)p.n_bd) {
            float pj[CPL], nv[CPL];
            Row<T, CPL>::load(zbd + j * p.ld_zbd + c0, pj);
            Row<T, CPL>::load(zbd + jn * p.ld_zbd + c0, nv);
            float s0 = 0.f, s1 = 0.f;
            if (p.sg_kind == SEGGER_LOSS_BCE) {
#pragma unroll
              for (int k = 0; k < CPL; ++k) { s0 = fmaf(a[k], pj[k], s0); s1 = fmaf(a[k], nv[k], s1); }
              s0 = lane_block_sum<16>(s0); s1 = lane_block_sum<16>(s1);
              const float dlp = (sigmoid_(s0) - 1.0f) * sc_sg, dln = sigmoid_(s1) * sc_sg;
#pragma unroll
              for (int k = 0; k < CPL; ++k) g[k] += dlp * pj[k] + dln * nv[k];
            } else {
              float dp[CPL], dn[CPL];
#pragma unroll
              for (int k = 0; k < CPL; ++k) {
                dp[k] = a[k] - pj[k] + p.sg_eps; dn[k] = a[k] - nv[k] + p.sg_eps;
                s0 = fmaf(dp[k], dp[k], s0); s1 = fmaf(dn[k], dn[k], s1);
              }
              s0 = lane_block_sum<16>(s0); s1 = lane_block_sum<16>(s1);
              const float dap = sqrtf(s0), dan = sqrtf(s1);
              if (dap - dan + p.sg_margin > 0.f) {
                const float ip_ = dap > 0.f ? sc_sg / dap : 0.f, in_ = dan > 0.f ? sc_sg / dan : 0.f;
#pragma unroll
                for (int k = 0; k < CPL; ++k) g[k] += dp[k] * ip_ - dn[k] * in_;
              }
            }
          }
        }
      }
      // (4) finish: a hot row (more contributions than its chain holds) through its accumulator, any other row directly
      if (total > kChainCap) hot_add(r, g);
      else finish_row(r, g);
    }
  } else if (blk < p.nb_tx + p.nb_bd) {
    // ---- loss_bd backward (csrc/heads.hip metric_kernel): own row, positive and negative rows by fp32 atomics
    const float sc = graw_of(p, 1);
    const int64_t base = (int64_t)(blk - p.nb_tx) * kItemsPerBlock;
#pragma unroll 1
    for (int q = wave * 4 + grp; q < kItemsPerBlock; q += 16) {
      const int64_t i = base + q;
      if (i >= p.n_bd) continue;
      const int64_t ip = p.bd_pos[i], in = p.bd_neg[i];
      const float w = p.bd_w[i];
      if ((uint64_t)ip >= (uint64_t)p.n_bd || (uint64_t)in >= (uint64_t)p.n_bd || w == 0.f) continue;
      float x[CPL], yp[CPL], yn[CPL];
      Row<T, CPL>::load(zbd + i * p.ld_zbd + c0, x);
      Row<T, CPL>::load(zbd + ip * p.ld_zbd + c0, yp);
      Row<T, CPL>::load(zbd + in * p.ld_zbd + c0, yn);
      float sxx = 0.f, spp = 0.f, snn = 0.f, sxp = 0.f, sxn = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        sxx = fmaf(x[k], x[k], sxx); spp = fmaf(yp[k], yp[k], spp); snn = fmaf(yn[k], yn[k], snn);
        sxp = fmaf(x[k], yp[k], sxp); sxn = fmaf(x[k], yn[k], sxn);
      }
      sxx = lane_block_sum<16>(sxx); spp = lane_block_sum<16>(spp); snn = lane_block_sum<16>(snn);
      sxp = lane_block_sum<16>(sxp); sxn = lane_block_sum<16>(sxn);
      const float nx = sqrtf(sxx), np_ = sqrtf(spp), nn_ = sqrtf(snn);
      const float cx = fmaxf(nx, p.bd_eps), cp = fmaxf(np_, p.bd_eps), cn = fmaxf(nn_, p.bd_eps);
      const float cos_p = sxp / (cx * cp), cos_n = sxn / (cx * cn);
      const float rp = cos_p - (1.f - p.bd_dpos[i]), rn = cos_n - (1.f - p.bd_dneg[i]);
      const float gp = 2.f * w * rp * sc, gn = 2.f * w * rn * sc;
      const float ax = (nx > p.bd_eps ? (gp * cos_p + gn * cos_n) / sxx : 0.f);
      const float bp = (np_ > p.bd_eps ? gp * cos_p / spp : 0.f), bn = (nn_ > p.bd_eps ? gn * cos_n / snn : 0.f);
      const float ip_ = gp / (cx * cp), in_ = gn / (cx * cn);
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        atomicAdd(p.gbd + i * C + c0 + k, ip_ * yp[k] + in_ * yn[k] - ax * x[k]);
        atomicAdd(p.gbd + ip * C + c0 + k, ip_ * x[k] - bp * yp[k]);
        atomicAdd(p.gbd + in * C + c0 + k, in_ * x[k] - bn * yn[k]);
      }
    }
  } else {
    // ---- loss_sg, boundary side: one workgroup per boundary j walks the triplets whose positive is j (csrc/heads.hip
    // triplet_grouped_kernel without the anchors' stores: the rows range above owns those).  Lane gl owns the channel
    // pairs 2 gl + 32 k: every atomic instruction lands on consecutive dwords.
    constexpr int KP = CPL / 2;
    const int64_t j = blk - p.nb_tx - p.nb_bd;
    if (j >= p.n_bd || p.n_sg <= 0) return;
    const int64_t beg = p.sg_indptr[j] + 4 * wave, end = p.sg_indptr[j + 1];
    if (beg >= end) return;                                       // wave-uniform
    const float sc = graw_of(p, 2) * (p.sg_kind == SEGGER_LOSS_BCE ? 0.5f : 1.0f) / (float)p.n_sg;
    float pj[KP][2], acc[KP][2];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      ld2(zbd + j * p.ld_zbd + 2 * gl + 32 * k, pj[k][0], pj[k][1]);
      acc[k][0] = 0.f; acc[k][1] = 0.f;
    }
    for (int64_t s0 = beg; s0 < end; s0 += 16) {
      const int64_t s = s0 + grp;
      bool ok = s < end;
      const int64_t e = ok ? (int64_t)p.sg_eid[s] : 0;
      int64_t ia = ok ? p.sg_src[e] : 0, in = ok ? p.sg_neg[e] : 0;
      if ((ok && p.sg_pos[e] != j) || (uint64_t)ia >= (uint64_t)p.n_tx || (uint64_t)in >= (uint64_t)p.n_bd) { ok = false; ia = in = 0; }
      float av[KP][2], nv[KP][2];
      float t0 = 0.f, t1 = 0.f;
      if (p.sg_kind == SEGGER_LOSS_BCE) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          ld2(ztx + ia * p.ld_ztx + 2 * gl + 32 * k, av[k][0], av[k][1]);
          ld2(zbd + in * p.ld_zbd + 2 * gl + 32 * k, nv[k][0], nv[k][1]);
          t0 = fmaf(av[k][0], pj[k][0], fmaf(av[k][1], pj[k][1], t0));
          t1 = fmaf(av[k][0], nv[k][0], fmaf(av[k][1], nv[k][1], t1));
        }
        t0 = lane_block_sum<16>(t0); t1 = lane_block_sum<16>(t1);
        if (ok) {
          const float dlp = (sigmoid_(t0) - 1.0f) * sc, dln = sigmoid_(t1) * sc;
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            acc[k][0] = fmaf(dlp, av[k][0], acc[k][0]); acc[k][1] = fmaf(dlp, av[k][1], acc[k][1]);
            atomicAdd(p.gbd + in * C + 2 * gl + 32 * k, dln * av[k][0]);
            atomicAdd(p.gbd + in * C + 2 * gl + 32 * k + 1, dln * av[k][1]);
          }
        }
      } else {
        float dp[KP][2], dn[KP][2];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          float n0, n1;
          ld2(ztx + ia * p.ld_ztx + 2 * gl + 32 * k, av[k][0], av[k][1]);
          ld2(zbd + in * p.ld_zbd + 2 * gl + 32 * k, n0, n1);
          dp[k][0] = av[k][0] - pj[k][0] + p.sg_eps; dp[k][1] = av[k][1] - pj[k][1] + p.sg_eps;
          dn[k][0] = av[k][0] - n0 + p.sg_eps;       dn[k][1] = av[k][1] - n1 + p.sg_eps;
          t0 = fmaf(dp[k][0], dp[k][0], t0); t0 = fmaf(dp[k][1], dp[k][1], t0);
          t1 = fmaf(dn[k][0], dn[k][0], t1); t1 = fmaf(dn[k][1], dn[k][1], t1);
        }
        t0 = lane_block_sum<16>(t0); t1 = lane_block_sum<16>(t1);
        const float dap = sqrtf(t0), dan = sqrtf(t1);
        if (ok && dap - dan + p.sg_margin > 0.f) {
          const float ip_ = dap > 0.f ? sc / dap : 0.f, in_ = dan > 0.f ? sc / dan : 0.f;
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            acc[k][0] -= dp[k][0] * ip_; acc[k][1] -= dp[k][1] * ip_;
            atomicAdd(p.gbd + in * C + 2 * gl + 32 * k, dn[k][0] * in_);
            atomicAdd(p.gbd + in * C + 2 * gl + 32 * k + 1, dn[k][1] * in_);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v = acc[k][q];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (grp == 0) atomicAdd(p.gbd + j * C + 2 * gl + 32 * k + q, v);
      }
    }
  }
}

int64_t blocks_of(int64_t n) { return (n + kItemsPerBlock - 1) / kItemsPerBlock; }

int fill_params(const segger_loss_head_args* a, bool bwd, LossHeadParams* out) {
  SEGGER_REQUIRE(a != nullptr, "segger_loss_head: args is NULL");
  SEGGER_REQUIRE(a->n_tx >= 0 && a->n_bd >= 0 && a->n_sg >= 0, "segger_loss_head: negative size");
  SEGGER_REQUIRE(a->channels == 32 || a->channels == 64 || a->channels == 128, "segger_loss_head: channels must be 32, 64 or 128");
  SEGGER_REQUIRE(a->dtype == SEGGER_F32 || a->dtype == SEGGER_BF16 || a->dtype == SEGGER_F16, "segger_loss_head: unknown dtype %d", a->dtype);
  const size_t es = a->dtype == SEGGER_F32 ? 4 : 2;
  const size_t row_align = (size_t)(a->channels / 16) * es >= 16 ? 16 : (size_t)(a->channels / 16) * es;
  SEGGER_REQUIRE(a->n_tx == 0 || (a->z_tx && ((uintptr_t)a->z_tx % 16) == 0 && ((size_t)a->ld_ztx * es) % row_align == 0 && a->ld_ztx >= a->channels),
                 "segger_loss_head: z_tx NULL, misaligned or ld < channels");
  SEGGER_REQUIRE(a->n_bd == 0 || (a->z_bd && ((uintptr_t)a->z_bd % 16) == 0 && ((size_t)a->ld_zbd * es) % row_align == 0 && a->ld_zbd >= a->channels),
                 "segger_loss_head: z_bd NULL, misaligned or ld < channels");
  SEGGER_REQUIRE(a->n_tx == 0 || (a->tx_pos && a->tx_neg), "segger_loss_head: NULL loss_tx triplets");
  SEGGER_REQUIRE(a->n_bd == 0 || (a->bd_pos && a->bd_neg && a->bd_dpos && a->bd_dneg && a->bd_w), "segger_loss_head: NULL loss_bd input");
  SEGGER_REQUIRE(a->n_sg == 0 || (a->sg_src && a->sg_pos && a->sg_neg), "segger_loss_head: NULL loss_sg triplets");
  SEGGER_REQUIRE(a->sg_kind == SEGGER_LOSS_TRIPLET || a->sg_kind == SEGGER_LOSS_BCE, "segger_loss_head: unknown sg_kind");
  SEGGER_REQUIRE(a->a && a->b && a->out, "segger_loss_head: NULL a / b / out");
  SEGGER_REQUIRE(a->n_tx < 0x3fffffffLL && a->n_bd < 0x3fffffffLL && a->n_sg < 0x7fffffffLL, "segger_loss_head: batch too large");
  const int64_t nb_tx = blocks_of(a->n_tx), nb_bd = blocks_of(a->n_bd);
  const int64_t nb_sg = bwd ? (a->n_sg > 0 ? a->n_bd : 0) : blocks_of(a->n_sg);
  SEGGER_REQUIRE(nb_tx + nb_bd + nb_sg < 0x7fffffffLL, "segger_loss_head: too many blocks");
  const size_t need = segger_loss_head_workspace_bytes(a->n_tx, a->n_bd, a->n_sg);
  if (!a->workspace || a->workspace_bytes < need) {
    set_error("segger_loss_head: workspace %zu < %zu bytes", a->workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  LossHeadParams p{};
  p.ztx = a->z_tx; p.ld_ztx = a->ld_ztx; p.n_tx = a->n_tx;
  p.zbd = a->z_bd; p.ld_zbd = a->ld_zbd; p.n_bd = a->n_bd;
  p.tx_pos = a->tx_pos; p.tx_neg = a->tx_neg; p.tx_margin = a->tx_margin; p.tx_eps = a->tx_eps;
  p.bd_pos = a->bd_pos; p.bd_neg = a->bd_neg; p.bd_dpos = a->bd_dpos; p.bd_dneg = a->bd_dneg; p.bd_w = a->bd_w; p.bd_eps = a->bd_eps;
  p.sg_src = a->sg_src; p.sg_pos = a->sg_pos; p.sg_neg = a->sg_neg; p.n_sg = a->n_sg;
  p.sg_margin = a->sg_margin; p.sg_eps = a->sg_eps; p.sg_kind = a->sg_kind;
  p.sg_indptr = a->sg_pos_indptr; p.sg_eid = a->sg_pos_eid; p.sg_of_tx = a->sg_of_tx;
  p.a = a->a; p.b = a->b; p.gout = a->grad_out; p.out = a->out; p.graw = a->grad_raw;
  p.tx_w = a->tx_w; p.state = a->tx_state; p.next = a->tx_next;
  p.hot_id = a->tx_hot_id; p.hot_acc = a->tx_hot_acc; p.max_hot = segger_loss_head_max_hot_rows(a->n_tx);
  p.ytx = a->y_tx; p.ld_ytx = a->ld_ytx; p.norm_eps = a->norm_eps;
  p.gtx = a->grad_tx; p.ld_gtx = a->ld_gtx; p.gbd = a->grad_bd;
  p.partial = static_cast<float*>(a->workspace); p.ticket = a->ticket;
  p.nb_tx = (int)nb_tx; p.nb_bd = (int)nb_bd; p.nb_sg = (int)nb_sg;
  p.defer = (a->reserved_ & SEGGER_LOSS_HEAD_DEFER_FINISH) ? 1 : 0;
  SEGGER_REQUIRE(!p.defer || a->grad_out, "segger_loss_head: DEFER_FINISH needs grad_out (the gradient the backward will receive)");
  p.nb_sg_fin = (int)blocks_of(a->n_sg);
  if (!bwd) {
    SEGGER_REQUIRE(!a->tx_w == !a->tx_state && !a->tx_state == !a->tx_next && !a->tx_next == !a->tx_hot_id && !a->tx_hot_id == !a->tx_hot_acc,
                   "segger_loss_head_fwd: tx_w, tx_state, tx_next, tx_hot_id and tx_hot_acc go together");
    SEGGER_REQUIRE(!a->grad_out || a->grad_raw, "segger_loss_head_fwd: grad_out needs grad_raw");
    SEGGER_REQUIRE(!a->grad_bd || ((uintptr_t)a->grad_bd % 16) == 0, "segger_loss_head_fwd: grad_bd must be 16-byte aligned");
  } else {
    SEGGER_REQUIRE(a->grad_raw && a->tx_w && a->tx_state && a->tx_next && a->tx_hot_id && a->tx_hot_acc,
                   "segger_loss_head_bwd: needs grad_raw and the forward's tx_w / tx_state / tx_next / tx_hot_id / tx_hot_acc");
    SEGGER_REQUIRE(a->n_tx == 0 || (a->grad_tx && ((uintptr_t)a->grad_tx % 16) == 0 && ((size_t)a->ld_gtx * es) % row_align == 0 && a->ld_gtx >= a->channels),
                   "segger_loss_head_bwd: grad_tx NULL, misaligned or ld < channels");
    SEGGER_REQUIRE(a->n_bd == 0 || a->grad_bd, "segger_loss_head_bwd: grad_bd is NULL");
    SEGGER_REQUIRE(!a->y_tx || (((uintptr_t)a->y_tx % 16) == 0 && ((size_t)a->ld_ytx * es) % row_align == 0 && a->ld_ytx >= a->channels),
                   "segger_loss_head_bwd: y_tx misaligned or ld < channels");
    SEGGER_REQUIRE(a->n_sg == 0 || (a->sg_pos_indptr && a->sg_pos_eid), "segger_loss_head_bwd: the segmentation triplets need their grouping by positive row");
  }
  *out = p;
  return SEGGER_OK;
}

template <bool BWD>
int launch(const segger_loss_head_args* a, hipStream_t stream) {
  LossHeadParams p;
  const int rc = fill_params(a, BWD, &p);
  if (rc != SEGGER_OK) return rc;
  const int64_t grid = (int64_t)p.nb_tx + p.nb_bd + p.nb_sg + ((BWD && p.defer) ? 1 : 0);
  if (grid == 0) return SEGGER_OK;
#define GO(T, CPL)                                                                                              \
  do {                                                                                                          \
    if (BWD) hipLaunchKernelGGL((loss_head_bwd_kernel<T, CPL>), dim3((unsigned)grid), dim3(256), 0, stream, p); \
    else     hipLaunchKernelGGL((loss_head_fwd_kernel<T, CPL>), dim3((unsigned)grid), dim3(256), 0, stream, p); \
  } while (0)
#define BY_C(T)                                       \
  switch (a->channels) {                              \
    case 32: GO(T, 2); break;                         \
    case 64: GO(T, 4); break;                         \
    default: GO(T, 8); break;                         \
  }
  switch (a->dtype) {
    case SEGGER_F32:  BY_C(float); break;
    case SEGGER_BF16: BY_C(bf16_t); break;
    default:          BY_C(f16_t); break;
  }
#undef BY_C
#undef GO
  SEGGER_LAUNCH_CHECK("loss_head kernel");
  if (!BWD && !p.defer) {
    hipLaunchKernelGGL(loss_head_finish_kernel, dim3(1), dim3(256), 0, stream, p);
    SEGGER_LAUNCH_CHECK("loss_head_finish_kernel");
  }
  return SEGGER_OK;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" size_t segger_loss_head_workspace_bytes(int64_t n_tx, int64_t n_bd, int64_t n_sg) {
  const int64_t nb = blocks_of(n_tx > 0 ? n_tx : 0) + blocks_of(n_bd > 0 ? n_bd : 0) + blocks_of(n_sg > 0 ? n_sg : 0);
  return (size_t)(nb > 0 ? nb : 1) * sizeof(float) + 16;
}

extern "C" int64_t segger_loss_head_max_hot_rows(int64_t n_tx) { return n_tx > 0 ? 2 * n_tx / (kChainCap + 1) + 2 : 0; }

extern "C" int segger_loss_head_supported(int32_t channels, int32_t dtype) {
  return (channels == 32 || channels == 64 || channels == 128) && (dtype == SEGGER_F32 || dtype == SEGGER_BF16 || dtype == SEGGER_F16);
}

extern "C" int segger_loss_head_fwd(const segger_loss_head_args* a, segger_stream_t stream) {
  return launch<false>(a, (hipStream_t)stream);
}

extern "C" int segger_loss_head_bwd(const segger_loss_head_args* a, segger_stream_t stream) {
  return launch<true>(a, (hipStream_t)stream);
}
