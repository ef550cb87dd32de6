// Fused GATv2 attention aggregation for gfx950: forward, destination-side
// backward and source-side backward.  See DESIGN.md "Kernels" for the layout.
//
// Geometry.  A feature row has H*C channels; every lane owns 8 consecutive
// channels (one 16-byte load for bf16/f16, two for f32), so a row is covered by
// LPR = H*C/8 lanes and a head by LPH = C/8 lanes.  Lanes are grouped in
// power-of-two groups of GS >= LPR lanes; a wave holds NG = 64/GS groups and
// therefore gathers NG neighbour rows per load instruction (flagship H=2,C=64:
// GS=16, 4 rows = 1 KiB per wave-instruction).
//   * group-per-row mode (WPR=false): each group owns one CSR row and walks its
//     edges; used when the average degree is small (tx-neighbors-tx, k~15).
//   * wave-per-row mode (WPR=true): the NG groups split one row's edges and
//     merge their online-softmax states at the end; used for high-degree rows
//     (tx-belongs-bd: tens to hundreds of transcripts per nucleus).
// Neighbour ids are fetched GS (or 64) at a time with one coalesced load and
// handed to the gathering lanes by DPP row-broadcast (GS=16) or ds_bpermute.
#pragma once
#include "common.h"

namespace segger {

// neighbour rows in flight per lane, per kernel
#ifndef SEGGER_FWD_UNROLL
#define SEGGER_FWD_UNROLL 4
#endif
#ifndef SEGGER_FWD_UNROLL_WPR
#define SEGGER_FWD_UNROLL_WPR 2
#endif
#ifndef SEGGER_DST_UNROLL
#define SEGGER_DST_UNROLL 4
#endif
#ifndef SEGGER_SRC_UNROLL
#define SEGGER_SRC_UNROLL 4
#endif

// minimum resident waves per SIMD the register allocator must leave room for
// (second __launch_bounds__ argument: 512 VGPRs / waves, granule 8)
#ifndef SEGGER_FWD_WAVES
#define SEGGER_FWD_WAVES 4
#endif
#ifndef SEGGER_BWD_DST_WAVES
#define SEGGER_BWD_DST_WAVES 3
#endif
#ifndef SEGGER_BWD_SRC_WAVES
#define SEGGER_BWD_SRC_WAVES 3
#endif

template <int H_, int LPH_>
struct Geo {
  static constexpr int H = H_;
  static constexpr int LPH = LPH_;
  static constexpr int C = LPH * 8;
  static constexpr int HC = H * C;
  static constexpr int LPR = H * LPH;
  static constexpr int GS = LPR <= 1 ? 1 : LPR <= 2 ? 2 : LPR <= 4 ? 4 : LPR <= 8 ? 8 : LPR <= 16 ? 16 : LPR <= 32 ? 32 : 64;
  static constexpr int NG = 64 / GS;
  static_assert(LPR <= 64, "row does not fit one wave");
};

struct GatParams {
  // graph (rows = the node this kernel iterates over)
  const int64_t* indptr;
  const int32_t* col;
  const int32_t* eid;
  const uint8_t* bits;           // nullable: per slot, bit h = dropout keep of head h (segger_dropout_bits): replaces the hash
  const int32_t* order;          // nullable: position -> row (degree-balanced visiting order, group-per-row mode)
  // block tables (segger_csr_block_tables; LDSG kernels): the distinct gathered rows of a workgroup's 16 positions
  const int32_t* blk_cnt; const int32_t* blk_src; const uint8_t* col_local;
  int64_t n_rows;
  int64_t n_edges;
  // features
  const void* xl; int64_t ld_xl;
  const void* xr; int64_t ld_xr;
  const float* att;
  const float* bias;
  // forward outputs
  void* out; int64_t ld_out;
  void* pre; int64_t ld_pre;     // fwd: output (nullable); bwd: input
  float* lse;                    // fwd: output (nullable); bwd: input
  float* alpha;
  // backward
  const void* gout; int64_t ld_go;
  void* gpre; int64_t ld_gp;     // dst pass: output; src pass: input
  float* dsum;                   // [n_dst, H, 2] = (lse, D) pairs -- dst pass: output; src pass: input
  void* zero_rows; int64_t ld_zero;   // src pass, optional: a [n_src, HC] matrix whose rows it zero-fills on the way
  void* gxl; int64_t ld_gxl;
  void* gxr; int64_t ld_gxr;
  float* slab;                   // [nblocks][2][HC] partial grad_att | grad_bias
  // scalars
  float slope;
  float drop_scale;              // 1/(1-p)
  uint32_t drop_thr;             // floor(p * 2^24); 0 = no dropout
  uint32_t seed_lo, seed_hi;      // halves of splitmix64(seed) when seed_dev is NULL
  const uint64_t* seed_dev;       // optional device word added to seed_raw before mixing (hipGraph replays)
  uint64_t seed_raw;
  int32_t apply_gelu;
  int32_t direct_gxl;            // dst pass: store grad_xl rows itself (sources with at most one out-edge)
  int32_t rows_per_wave_iter;    // dst pass: row batches each wave walks
  int64_t nblocks, nblocks_padded;
};

// ---- neighbour-id hand-off -------------------------------------------------
// One coalesced load fetches GS (group-per-row) or 64 (wave-per-row) neighbour
// ids; lane `slot` holds the id every lane of the group needs next.
//  * GS == 16, group-per-row: the group is one DPP row.  The ids of the next U
//    edges always sit in lanes 0..U-1 of the row (row_newbcast:u) because the
//    id register is rotated by U lanes (row_ror:U) after every batch.
//  * otherwise: ds_bpermute with a computed slot.
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// The kernel bodies are functions (two of them can share a launch: gatv2_*_pair_kernel) and take the parameter block BY
// VALUE: handed over by const reference the fp32 source pass came out of the register allocator 32 % slower (2.70 vs
// 2.04 ms for the C2 backward pair, same instruction mix; tools/build_variant.sh A/B), by value all three storage types
// match the bodies written directly into the kernels.
#ifndef SEGGER_BODY_PARAM
#define SEGGER_BODY_PARAM const GatParams
#endif
constexpr int kDppRowRor0 = 0x120;   // row_ror:n

// Row walker shared by the three kernels.  Calls body(valid[U], nbr[U], eid[U])
// for successive batches of U edges of this group's share of row [beg,end).
// All lanes of a group see identical arguments.  In wave-per-row mode the loop
// is wave-uniform and a group may be handed a batch with valid[0] == false.
// per-edge side information handed to the body next to the neighbour id
constexpr int kMetaNone = 0, kMetaEid = 1, kMetaBits = 2;
// The lane id recomputed where it is used (two VALU instructions per 64-edge chunk).  In wave-per-row mode the chunk loop
// reads `lane` once per chunk for its coalesced id load; kept live across the edge loops it is the value the register
// allocator picks to SPILL at 4 waves per SIMD (as a zero-extended 64-bit pair, reloaded behind an s_waitcnt vmcnt(0) at
// every chunk head: tools/kernel_resources.py).  `volatile`: LLVM may not hoist it back out of the loop.
__device__ __forceinline__ int lane_id_fresh() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  return x;
}
template <int GS, bool WPR, int META, int U, typename Body>
__device__ __forceinline__ void walk_row(const int32_t* __restrict__ col, const void* __restrict__ meta,
                                         int64_t beg, int64_t end, int lane, int grp, int gl, Body&& body) {
  constexpr bool NEED_EID = META != kMetaNone;
  const int32_t* __restrict__ eid = static_cast<const int32_t*>(meta);
  const uint8_t* __restrict__ bits = static_cast<const uint8_t*>(meta);
  constexpr int NG = 64 / GS;
  constexpr int CHUNK = WPR ? 64 : GS;      // ids fetched per coalesced load (per wave / per group)
  constexpr bool kDpp = !WPR && GS == 16;
  static_assert(GS % U == 0, "group size must be a multiple of the edge unroll");
  for (int64_t e0 = beg; e0 < end; e0 += CHUNK) {
    const int me = WPR ? lane_id_fresh() : gl;
    // slots of this chunk that hold an edge (32-bit: the per-slot validity tests are one v_cmp_lt_i32 each instead of a
    // 64-bit add + compare)
    const int rem = (int)(end - e0 < (int64_t)CHUNK ? end - e0 : (int64_t)CHUNK);
    int myc = 0, myeid = 0;
    if (me < rem) {
      myc = col[e0 + me];
      if constexpr (META == kMetaEid) myeid = eid[e0 + me];
      if constexpr (META == kMetaBits) myeid = bits[e0 + me];
    }
#pragma unroll 1
    for (int t = 0; t < GS; t += U) {
      // wave-per-row: uniform over the wave (end is); group-per-row: uniform over the group,
      // and every id a group reads lives in its own lanes
      if ((WPR ? t * NG : t) >= rem) break;
      bool valid[U];
      int nbr[U], ed[U];
      static_for<U>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value;
        const int T = t + u;
        valid[u] = (WPR ? T * NG + grp : T) < rem;
        if constexpr (kDpp) {
          nbr[u] = dpp_i<kDppRowBcast0 + u>(myc);
          ed[u] = NEED_EID ? dpp_i<kDppRowBcast0 + u>(myeid) : 0;
        } else {
          const int slot = WPR ? (T * NG + grp) : (grp * GS + T);
          nbr[u] = __builtin_amdgcn_ds_bpermute(slot << 2, myc);
          ed[u] = NEED_EID ? __builtin_amdgcn_ds_bpermute(slot << 2, myeid) : 0;
        }
      });
      body(std::integral_constant<int, META>{}, valid, nbr, ed);
      if constexpr (kDpp) {
        myc = dpp_i<kDppRowRor0 + 16 - U>(myc);   // lane i <- lane i+U
        if constexpr (NEED_EID) myeid = dpp_i<kDppRowRor0 + 16 - U>(myeid);
      }
    }
  }
}

// The same walk with the gathers of batch t + 1 issued BEFORE the arithmetic of batch t (two register buffers, the
// batch loop unrolled by two so that the buffer index is a compile-time constant).  issue(buf, nbr[U]) starts the
// gathers of a batch into buffer `buf`; compute(buf, meta, valid[U], ed[U]) consumes it.  The prefetch is
// unconditional straight-line code: behind the last batch of a chunk it fetches rows the batch does not use (ids past
// the row's end are 0 or ids of the same chunk: valid rows, cache hits) -- a conditional load would be an event the
// compiler's wait-count analysis cannot count on, and the wait for the current buffer would become vmcnt(0), i.e. wait
// for the prefetch as well.  Why: every row has the same length, so the waves of a CU run in phase -- all issue their
// gathers, all wait for the texture addresser to work through them (~16 cycles per 1-KB gather, ~1000 cycles per
// round and CU), all compute.  With the gathers compiled out the forward takes 0.50 ms instead of 0.66, with every
// gather an L1 hit (`GRAPH=band`) still 0.655: it is this serialisation, not latency, that the prefetch overlaps.
template <int GS, bool WPR, int META, int U, typename ColT, typename Issue, typename Compute>
__device__ __forceinline__ void walk_row_prefetch(const ColT* __restrict__ col, const void* __restrict__ meta,
                                                  int64_t beg, int64_t end, int lane, int grp, int gl,
                                                  Issue&& issue, Compute&& compute) {
  constexpr bool NEED_EID = META != kMetaNone;
  const int32_t* __restrict__ eid = static_cast<const int32_t*>(meta);
  const uint8_t* __restrict__ bits = static_cast<const uint8_t*>(meta);
  constexpr int NG = 64 / GS;
  constexpr int CHUNK = WPR ? 64 : GS;
  constexpr bool kDpp = !WPR && GS == 16 && 2 * U <= 16;
  static_assert(GS % U == 0, "group size must be a multiple of the edge unroll");
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  for (int64_t e0 = beg; e0 < end; e0 += CHUNK) {
    const int me = WPR ? lane_id_fresh() : gl;
    const int rem = (int)(end - e0 < (int64_t)CHUNK ? end - e0 : (int64_t)CHUNK);
    int myc = 0, myeid = 0;
    if (me < rem) {
      myc = col[e0 + me];
      if constexpr (META == kMetaEid) myeid = eid[e0 + me];
      if constexpr (META == kMetaBits) myeid = bits[e0 + me];
    }
    // ids of the batch `ahead` (0 or 1) batches behind position t of the chunk (DPP: the id register has been rotated
    // by t lanes, so the batch at t + ahead * U sits in lanes ahead * U .. ahead * U + U - 1 of every row)
    auto ids = [&](auto ahead_c, int t, int (&nbr)[U]) {
      constexpr int ahead = decltype(ahead_c)::value;
      static_for<U>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value;
        if constexpr (kDpp) {
          nbr[u] = dpp_i<kDppRowBcast0 + ahead * U + u>(myc);
        } else {
          const int T = t + ahead * U + u;
          const int slot = WPR ? (T * NG + grp) : (grp * GS + T);
          nbr[u] = __builtin_amdgcn_ds_bpermute(slot << 2, myc);      // (slot taken mod 64: some lane's id or 0)
        }
      });
    };
    auto batch = [&](auto buf_c, int t) {                 // validity / per-edge side information of the batch at t, then its arithmetic
      bool valid[U];
      int ed[U];
      static_for<U>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value;
        const int T = t + u;
        valid[u] = (WPR ? T * NG + grp : T) < rem;
        if constexpr (kDpp) {
          ed[u] = NEED_EID ? dpp_i<kDppRowBcast0 + u>(myeid) : 0;
        } else {
          const int slot = WPR ? (T * NG + grp) : (grp * GS + T);
          ed[u] = NEED_EID ? __builtin_amdgcn_ds_bpermute(slot << 2, myeid) : 0;
        }
      });
      compute(buf_c, std::integral_constant<int, META>{}, valid, ed);
      if constexpr (kDpp) {
        myc = dpp_i<kDppRowRor0 + 16 - U>(myc);           // lane i <- lane i+U
        if constexpr (NEED_EID) myeid = dpp_i<kDppRowRor0 + 16 - U>(myeid);
      }
    };
    // the chunk's ids / side bytes must have landed before the loop: left pending, the loop head (which merges "id loads
    // newest" from the entry with "prefetch newest" from the back edge) waits vmcnt(0) in every iteration
    asm volatile("" ::"v"(myc), "v"(myeid));
    int nb[U];
    ids(B0{}, 0, nb);
    issue(B0{}, nb);
#pragma unroll 1
    for (int t = 0; ; t += 2 * U) {
      // sched_barrier: the machine scheduler otherwise sinks the prefetch below the arithmetic it is meant to overlap
      ids(B1{}, t, nb);                                   // batch t + U (rows it may not need: see above)
      issue(B1{}, nb);
      __builtin_amdgcn_sched_barrier(0);
      batch(B0{}, t);
      __builtin_amdgcn_sched_barrier(0);
      if ((WPR ? (t + U) * NG : t + U) >= rem) break;     // group-uniform (wave-uniform in wave-per-row mode)
      ids(B1{}, t + U, nb);                               // batch t + 2U
      issue(B0{}, nb);
      __builtin_amdgcn_sched_barrier(0);
      batch(B1{}, t + U);
      __builtin_amdgcn_sched_barrier(0);
      if ((WPR ? (t + 2 * U) * NG : t + 2 * U) >= rem) break;
    }
  }
}

// ----------------------------------------------------------------------------
// Arithmetic layout.  The kernels are VALU-bound (rocprofv3: VALU ~95 % busy,
// HBM traffic 1/4 of the algorithmic bytes), so the inner loops are written for
// instruction count:
//  * 8 channels per lane live as 4 f32x2 pairs -> v_pk_add_f32 / v_pk_fma_f32;
//  * leaky_relu(t) = c1*t + c2*|t| with c1=(1+slope)/2, c2=(1-slope)/2, so the
//    logit  sum_k att_k*lrelu(t_k) = sum_k a1_k*t_k + a2_k*|t_k|  is one packed
//    fma (a1 = att*c1*log2e) plus one fma with the free |.| source modifier
//    (a2 = att*c2*log2e) per channel -- no multiply + max;
//  * lrelu'(t) = c1 + c2*sgn(t) (sgn(0) = -1 as torch's `t > 0 ? 1 : slope`), so
//    sum_e de*att*lrelu'(t) = att*(c1*sum_e de + c2*sum_e de*sgn(t)): only the
//    sign sum is per channel; likewise grad_att = c1*sum de*t + c2*sum de*|t|.
// ----------------------------------------------------------------------------
template <typename T> struct Raw8 {
  u32x4 r;
  __device__ __forceinline__ void load(const void* p) { r = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void get(f32x2 (&f)[4]) const;
};
template <> __device__ __forceinline__ void Raw8<bf16_t>::get(f32x2 (&f)[4]) const {
  f[0] = f32x2{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
  f[1] = f32x2{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
  f[2] = f32x2{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u)};
  f[3] = f32x2{__uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
template <> __device__ __forceinline__ void Raw8<f16_t>::get(f32x2 (&f)[4]) const {
  float a, b;
  Vec8<f16_t>::unpack(r.x, a, b); f[0] = f32x2{a, b};
  Vec8<f16_t>::unpack(r.y, a, b); f[1] = f32x2{a, b};
  Vec8<f16_t>::unpack(r.z, a, b); f[2] = f32x2{a, b};
  Vec8<f16_t>::unpack(r.w, a, b); f[3] = f32x2{a, b};
}
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const void* p) {
    a = *reinterpret_cast<const f32x4*>(p); b = *(reinterpret_cast<const f32x4*>(p) + 1);
  }
  __device__ __forceinline__ void get(f32x2 (&f)[4]) const {
    f[0] = f32x2{a.x, a.y}; f[1] = f32x2{a.z, a.w}; f[2] = f32x2{b.x, b.y}; f[3] = f32x2{b.z, b.w};
  }
};

template <typename T>
__device__ __forceinline__ void load_pairs(const T* p, f32x2 (&f)[4]) {
  Raw8<T> r; r.load(p); r.get(f);
}
template <typename T>
__device__ __forceinline__ void store_pairs(T* p, const f32x2 (&f)[4]) {
  const float v[8] = {f[0].x, f[0].y, f[1].x, f[1].y, f[2].x, f[2].y, f[3].x, f[3].y};
  Vec8<T>::store(p, v);
}

#ifdef SEGGER_SCALAR_PAIRS
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  return f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}
#else
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }

// gathered row address: base (already offset to this lane's channels) + id * row-bytes  (one v_mad_u64_u32)
__device__ __forceinline__ const void* row_ptr(const char* base, int id, uint32_t ld_bytes) {
  return base + (uint64_t)(uint32_t)id * ld_bytes;
}

// sum_k a1_k*t_k + a2_k*|t_k| over this lane's 8 channels (t = v + xr), before the head reduction
__device__ __forceinline__ float logit_partial(const f32x2 (&t)[4], const f32x2 (&a1)[4], const f32x2 (&a2)[4]) {
  f32x2 pp = f32x2{0.f, 0.f};
  float pa = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pp = pk_fma(a1[i], t[i], pp);
    pa = __builtin_fmaf(a2[i].x, __builtin_fabsf(t[i].x), pa);
    pa = __builtin_fmaf(a2[i].y, __builtin_fabsf(t[i].y), pa);
  }
  return pp.x + pp.y + pa;
}

// ---- sign algebra of the backward passes ------------------------------------
// Both passes carry nt = -(x_l + x_r) instead of t, formed as (-a) - b with every zero of the PER-ROW operand a
// canonicalised to -0: then nt is +0 whenever t is +-0 and sign(nt) == (t > 0) exactly (torch's leaky_relu
// derivative is `t > 0 ? 1 : slope`, so sgn(+-0) = -1).  de * sgn(t) is then one bit operation,
//   s = bits(-de) ^ (bits(nt) & 0x80000000)                     (v_bitop3_b32),
// which also yields de * |t| = s * t without a compare / select per channel.
__device__ __forceinline__ f32x2 neg_canon_zero(f32x2 a) {          // -a, with +-0 -> +0
  // IEEE round-to-nearest: (+0) - (+0) = +0 and (+0) - (-0) = +0, and 0 - a = -a otherwise: one subtraction instead
  // of a compare + select per channel (the compiler may not fold 0 - a into -a: it is not compiled with nsz)
  return f32x2{0.f, 0.f} - a;
}
__device__ __forceinline__ float sign_mul(float nde, float nt) {     // de * sgn(t) given -de and nt
  // a ^ (b & c) as one v_bitop3_b32 (truth table 0xF0 ^ (0xCC & 0xAA) = 0x78); the compiler does not fold the
  // and + xor pair by itself
  return __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(nde), __float_as_uint(nt), 0x80000000u, 0x78));
}

// position -> row through the optional degree-balanced order (include/segger_amd.h: segger_csr_row_order)
// -> the row as a 32-bit id (node ids are int32 throughout: `col`): one register instead of two across the edge walk --
// the 64-bit row was what the allocator spilled at 4 waves per SIMD -- and in wave-per-row mode, where every lane of the
// wave works on the same row, a SCALAR register (readfirstlane).  Addresses are formed as (int64_t)row * ld where used.
template <bool WPR>
__device__ __forceinline__ int visit_row(const int32_t* __restrict__ order, int64_t pos, bool ok) {
  if constexpr (WPR) return __builtin_amdgcn_readfirstlane((int)pos);
  return (order != nullptr && ok) ? order[pos] : (int)pos;
}

struct LaneGeo {
  int lane, wave, grp, gl, h, ch0;
  bool lane_on;
};
template <typename G>
__device__ __forceinline__ LaneGeo lane_geo() {
  LaneGeo g;
  g.lane = threadIdx.x & 63; g.wave = threadIdx.x >> 6;
  g.grp = g.lane / G::GS; g.gl = g.lane % G::GS;
  g.lane_on = g.gl < G::LPR;                 // GS > LPR when H*C/8 is not a power of two
  g.h = g.lane_on ? g.gl / G::LPH : 0;       // idle lanes shadow lane 0 and never store
  g.ch0 = g.lane_on ? g.gl * 8 : 0;
  return g;
}

// att -> (a1, a2) = att * {c1, c2} * scale
__device__ __forceinline__ void load_att(const float* att, int ch0, float slope, float scale, f32x2 (&a1)[4], f32x2 (&a2)[4]) {
  const float c1 = 0.5f * (1.f + slope) * scale, c2 = 0.5f * (1.f - slope) * scale;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 a = f32x2{att[ch0 + 2 * i], att[ch0 + 2 * i + 1]};
    a1[i] = a * c1; a2[i] = a * c2;
  }
}

// ============================================================================
// Forward
// ============================================================================
// LDSG (group-per-row, 16-bit rows of exactly 16 lanes): the workgroup's 16 rows gather from the <= SEGGER_BLOCK_CAP DISTINCT
// source rows listed in its block table.  They are loaded once, coalesced, into LDS (ceil(cnt / 16) row loads per 16-lane
// loader instead of one per edge: 58 instead of 240 on the C2 tile) and the edge walk gathers with ds_read_b128 from slot
// numbers (col_local) -- no per-edge work for the texture addresser, a 32-bit LDS address instead of a 64-bit pointer.
template <typename T, int H, int LPH, bool WPR, bool LDSG = false>
__device__ __forceinline__ void gatv2_fwd_body(SEGGER_BODY_PARAM p, int64_t bid) {
  using G = Geo<H, LPH>;
  static_assert(!LDSG || (!WPR && G::LPR == 16 && sizeof(T) == 2), "LDS gathers: group-per-row, 256-byte rows");
  constexpr int ROWB = G::HC * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char tile[LDSG ? SEGGER_BLOCK_CAP * ROWB : 16];
  // wave-per-row: 2 rows in flight per group (NG groups: 2 NG edges per wave batch) -- the merge of the groups' states and
  // the wave-wide id hand-off need the registers the second pair of row buffers would take (4 in flight spills at 4 waves
  // per SIMD; these rows are the few-thousand-block tx-belongs-bd side, latency-bound beside the tx-neighbors-tx blocks)
  constexpr int UF = WPR ? SEGGER_FWD_UNROLL_WPR : SEGGER_FWD_UNROLL;
  constexpr int GS = G::GS, NG = G::NG, U = UF < GS ? UF : GS;
  const int64_t blk = xcd_remap(bid, p.nblocks_padded, p.nblocks);
  if (blk < 0) return;
  const LaneGeo L = lane_geo<G>();
  const int h = L.h, ch0 = L.ch0;
  const bool head_leader = L.lane_on && (L.gl % LPH) == 0;
  const int64_t pos = WPR ? (blk * 4 + L.wave) : ((blk * 4 + L.wave) * NG + L.grp);
  const bool row_ok = pos < p.n_rows;
  const int row = visit_row<WPR>(p.order, pos, row_ok);
  const char* __restrict__ xl = static_cast<const char*>(p.xl) + (size_t)ch0 * sizeof(T);
  const uint32_t ld_xl = (uint32_t)(p.ld_xl * sizeof(T));
  const bool dropout = p.drop_thr != 0;
  uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
  if (dropout && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    seed_lo = (uint32_t)mixed; seed_hi = (uint32_t)(mixed >> 32);
  }
  const bool want_alpha = p.alpha != nullptr;

  f32x2 a1[4], a2[4], xr[4], acc[4];
  load_att(p.att, ch0, p.slope, kLog2e, a1, a2);      // logits come out in base-2 units
#pragma unroll
  for (int i = 0; i < 4; ++i) { xr[i] = splat(0.f); acc[i] = splat(0.f); }
  int64_t beg = 0, end = 0;
  if (row_ok) {
    beg = p.indptr[row];
    end = p.indptr[row + 1];
    load_pairs(static_cast<const T*>(p.xr) + row * p.ld_xr + ch0, xr);
  }

  float m = -INFINITY, s = 0.f;                       // online softmax state

  if constexpr (LDSG) {
    // stage the block's distinct source rows: 16 loaders of 16 lanes, one 256-byte row each per round
    const int cnt = p.blk_cnt[blk];
    const int32_t* __restrict__ srcs = p.blk_src + blk * SEGGER_BLOCK_CAP;
    const int ldr = threadIdx.x >> 4, piece = (threadIdx.x & 15) * 16;
    // all of a loader's ids first, then all of its rows, then the LDS writes: two memory latencies per workgroup, not two per
    // round (slots past `cnt` load row 0 and are not written)
    constexpr int ROUNDS = SEGGER_BLOCK_CAP / 16;
    int sid[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) sid[k] = ldr + 16 * k < cnt ? srcs[ldr + 16 * k] : 0;
    u32x4 sv[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k)
      sv[k] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.xl) + (uint64_t)(uint32_t)sid[k] * ld_xl + piece);
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k)
      if (ldr + 16 * k < cnt) *reinterpret_cast<u32x4*>(tile + (ldr + 16 * k) * ROWB + piece) = sv[k];
    __syncthreads();
  }
  const uint32_t lds_lane = (uint32_t)ch0 * (uint32_t)sizeof(T);

  const int hbit = 1 << h;
  Raw8<T> rawbuf[2][U];                               // gathered rows of the batch in work and of the next one
  auto issue = [&](auto buf_c, const int (&nbr)[U]) {
    constexpr int B = decltype(buf_c)::value;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (LDSG) rawbuf[B][u].r = *reinterpret_cast<const u32x4*>(tile + (uint32_t)nbr[u] * ROWB + lds_lane);
      else rawbuf[B][u].load(row_ptr(xl, nbr[u], ld_xl));   // invalid slots read row 0 / a row of the chunk
    }
  };
  auto body = [&](auto buf_c, auto meta_c, const bool (&valid)[U], const int (&ed)[U]) {
    constexpr int META = decltype(meta_c)::value;
    constexpr int B = decltype(buf_c)::value;
    if (!valid[0]) return;                            // wave-per-row tail (group-uniform)
    Raw8<T> (&raw)[U] = rawbuf[B];
    float e[U];
    float mx = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x2 v[4], t[4];
      raw[u].get(v);
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = v[i] + xr[i];
      const float pl = lane_block_sum<LPH>(logit_partial(t, a1, a2));
      e[u] = valid[u] ? pl : -INFINITY;
      mx = fmaxf(mx, e[u]);
    }
    const float sc = fast_exp2(m - mx);               // m = -inf -> 0 ; mx finite because valid[0]
    s *= sc;
    const f32x2 sc2 = splat(sc);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = acc[i] * sc2;
    m = mx;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pe = fast_exp2(e[u] - mx);          // invalid -> 0
      s += pe;
      float w = pe;
      if constexpr (META == kMetaBits) {
        w = (ed[u] & hbit) ? pe * p.drop_scale : 0.f;
      } else if constexpr (META == kMetaEid) {
        if (dropout) w = dropout_keep((uint32_t)ed[u], H, h, seed_lo, seed_hi, p.drop_thr) ? pe * p.drop_scale : 0.f;
        if (want_alpha && valid[u] && head_leader) p.alpha[(int64_t)ed[u] * H + h] = e[u];
      }
      f32x2 v[4];
      raw[u].get(v);
      const f32x2 w2 = splat(w);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = pk_fma(w2, v[i], acc[i]);
    }
  };
  // 16-bit storage: the next batch's rows are requested before this batch's arithmetic.  fp32 rows take twice the
  // registers (the second buffer spills 20-65 of them at 4 waves per SIMD): plain walk, one buffer.
  auto plain = [&](auto meta_c, const bool (&valid)[U], const int (&nbr)[U], const int (&ed)[U]) {
    issue(std::integral_constant<int, 0>{}, nbr);
    body(std::integral_constant<int, 0>{}, meta_c, valid, ed);
  };
  auto walk = [&](auto meta_c, const void* meta) {
    constexpr int META = decltype(meta_c)::value;
    if constexpr (LDSG) walk_row_prefetch<GS, WPR, META, U>(p.col_local, meta, beg, end, L.lane, L.grp, L.gl, issue, body);
    else if constexpr (sizeof(T) == 2) walk_row_prefetch<GS, WPR, META, U>(p.col, meta, beg, end, L.lane, L.grp, L.gl, issue, body);
    else walk_row<GS, WPR, META, U>(p.col, meta, beg, end, L.lane, L.grp, L.gl, plain);
  };
  if (want_alpha || (dropout && !p.bits)) walk(std::integral_constant<int, kMetaEid>{}, p.eid);
  else if (dropout) walk(std::integral_constant<int, kMetaBits>{}, p.bits);
  else walk(std::integral_constant<int, kMetaNone>{}, nullptr);

  if constexpr (WPR) {
    // merge the NG groups' online-softmax states
#pragma unroll
    for (int off = GS; off < 64; off <<= 1) {
      const float m_o = __shfl_xor(m, off, 64);
      const float s_o = __shfl_xor(s, off, 64);
      const float mn = fmaxf(m, m_o);
      const float a = (m == mn) ? 1.f : fast_exp2(m - mn);
      const float b = (m_o == mn) ? 1.f : fast_exp2(m_o - mn);
      s = s * a + s_o * b;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 o = f32x2{__shfl_xor(acc[i].x, off, 64), __shfl_xor(acc[i].y, off, 64)};
        acc[i] = acc[i] * a + o * b;
      }
      m = mn;
    }
  }

  const float lse = m + fast_log2(s);                 // -inf for a destination without in-edges
  const float inv = s > 0.f ? fast_rcp(s) : 0.f;      // (v_rcp_f32: 1 ulp; the IEEE division is ten instructions per row)
  // fp32 rows: the epilogue's lane offset recomputed from a fresh lane id -- kept live across the edge walk as a
  // zero-extended 64-bit pair it was what the allocator spilled at 4 waves per SIMD (12-20 B of scratch in the
  // <float, 2, 8> forward kernels; outside the edge loops, but scratch all the same: tests/test_kernel_resources.py)
  // ... and so was the row id (a sign-extended 64-bit pair in group-per-row mode): re-derived from the scalar block /
  // wave position and the fresh lane id (one more read of order[] where a visiting order is attached)
  int ch0e = L.ch0, rowe = row;
  if constexpr (sizeof(T) == 4) {
    const int lane_f = lane_id_fresh();
    ch0e = L.lane_on ? (lane_f % GS) * 8 : 0;
    if constexpr (!WPR) {
      const int64_t pos_f = (blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * NG + lane_f / GS;
      rowe = visit_row<false>(p.order, pos_f, pos_f < p.n_rows);
    }
  }
  if (row_ok && L.lane_on && (!WPR || L.grp == 0)) {
    const int ch0 = ch0e, row = rowe;                 // (shadow the function-scope values)
    f32x2 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = acc[i] * inv;
      if (p.bias) o[i] = o[i] + f32x2{p.bias[ch0 + 2 * i], p.bias[ch0 + 2 * i + 1]};
    }
    if (p.pre && (p.pre != p.out || p.apply_gelu))
      store_pairs(static_cast<T*>(p.pre) + row * p.ld_pre + ch0, o);
    if (p.apply_gelu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = f32x2{gelu_erf(o[i].x), gelu_erf(o[i].y)};
    }
    store_pairs(static_cast<T*>(p.out) + row * p.ld_out + ch0, o);
    if (p.lse && head_leader) p.lse[(int64_t)row * H + h] = lse;
#ifdef EXP_EXTRA_STORES      // bounding build: the bytes a fused next-layer projection would write from here
    if (p.gxl) {
      T* q = static_cast<T*>(p.gxl) + row * p.ld_gxl + ch0;
      store_pairs(q, o); store_pairs(q + H * LPH * 8, o); store_pairs(q + 2 * H * LPH * 8, o);
    }
#endif
  }

  if (want_alpha && row_ok && head_leader) {
    // second sweep over this row's edges: logits -> normalised (dropped-out) coefficients.
    // Every lane re-reads exactly the entries it stored itself in the main loop
    // (edge (e - beg) % NG == grp in wave-per-row mode), so program order suffices.
    const int step = WPR ? NG : 1;
    for (int64_t e = beg + (WPR ? L.grp : 0); e < end; e += step) {
      const int64_t id = p.eid[e];
      float a = fast_exp2(p.alpha[id * H + h] - lse);
      if (dropout) a = dropout_keep((uint32_t)id, H, h, seed_lo, seed_hi, p.drop_thr) ? a * p.drop_scale : 0.f;
      p.alpha[id * H + h] = a;
    }
  }
}

template <typename T, int H, int LPH, bool WPR>
__global__ __launch_bounds__(256, SEGGER_FWD_WAVES) void gatv2_fwd_kernel(GatParams p) {
#ifdef EXP_PERSIST           // bounding build: EXP_PERSIST workgroups per CU walk their XCD's row blocks in a loop
  const int64_t per = p.nblocks_padded / kNumXcd, k = gridDim.x / kNumXcd;
  for (int64_t i = blockIdx.x / kNumXcd; i < per; i += k) gatv2_fwd_body<T, H, LPH, WPR>(p, i * kNumXcd + blockIdx.x % kNumXcd);
#else
  gatv2_fwd_body<T, H, LPH, WPR>(p, blockIdx.x);
#endif
}

template <typename T, int H, int LPH>
__global__ __launch_bounds__(256, SEGGER_FWD_WAVES) void gatv2_fwd_lds_kernel(GatParams p) {
  gatv2_fwd_body<T, H, LPH, false, true>(p, blockIdx.x);
}

// Two edge types of one hetero layer in ONE launch: the blocks of `b` (wave-per-row: tx-belongs-bd, a few thousand short
// latency-bound blocks) are dispatched first and run beside the blocks of `a` (group-per-row: tx-neighbors-tx) instead
// of as a 30 us launch of their own.  b.nblocks_padded is a multiple of the XCD count, so the XCD-contiguous block map of
// `a` is unchanged.
template <typename T, int H, int LPH>
__global__ __launch_bounds__(256, SEGGER_FWD_WAVES) void gatv2_fwd_pair_kernel(GatParams a, GatParams b) {
  if ((int64_t)blockIdx.x < b.nblocks_padded) gatv2_fwd_body<T, H, LPH, true>(b, blockIdx.x);
  else gatv2_fwd_body<T, H, LPH, false>(a, (int64_t)blockIdx.x - b.nblocks_padded);
}

// ============================================================================
// Backward, destination side:  grad_pre, dsum, grad_xr, partial grad_att / grad_bias
//   g      = grad_out * gelu'(pre)                    (or grad_out)
//   D[j,h] = sum_c g[j,h,c] * (pre[j,h,c] - bias)     = sum_i a_ij * dL/da_ij
//   de_ij  = a_ij * (keep/(1-p) * <g_j, x_l[i]>_h - D[j,h])
//   grad_xr[j] = att * (c1 * sum_i de_ij + c2 * sum_i de_ij * sgn(t_ij))
//   grad_att   = c1 * sum_ij de_ij * t_ij + c2 * sum_ij de_ij * |t_ij| ;  grad_bias = sum_j g[j]
// ============================================================================
// DIRECT: every source node has at most one out-edge (segger's tx-belongs-bd: a transcript lies in at most one
// boundary), so grad_xl[i] has a single term and this pass stores it itself -- no by-source view, no source pass.
// The caller zero-fills grad_xl first (sources without an out-edge).
// (the direct form keeps 8 more values per lane live: it is built for 2 waves per SIMD instead of spilling)
template <typename T, int H, int LPH, bool WPR, bool DIRECT, int UFORCE = 0>
__device__ __forceinline__ void gatv2_bwd_dst_body(SEGGER_BODY_PARAM p, int64_t bid) {
  using G = Geo<H, LPH>;
  // 4 rows in flight measured -2.6 % on the flagship geometry (H = 2, group-per-row); the other variants would spill
  constexpr int UD = UFORCE ? UFORCE : ((!WPR && H == 2) || DIRECT) ? SEGGER_DST_UNROLL : 2;
  constexpr int GS = G::GS, NG = G::NG, U = UD < GS ? UD : GS, HC = G::HC;
  __shared__ float red[4][2][HC];
  const int64_t blk = xcd_remap(bid, p.nblocks_padded, p.nblocks);
  if (blk < 0) return;
  const LaneGeo L = lane_geo<G>();
  const int h = L.h, ch0 = L.ch0;
  const bool head_leader = L.lane_on && (L.gl % LPH) == 0;
  const char* __restrict__ xl = static_cast<const char*>(p.xl) + (size_t)ch0 * sizeof(T);
  const uint32_t ld_xl = (uint32_t)(p.ld_xl * sizeof(T));
  const bool dropout = p.drop_thr != 0;
  uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
  if (dropout && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    seed_lo = (uint32_t)mixed; seed_hi = (uint32_t)(mixed >> 32);
  }
  constexpr int RPW = WPR ? 1 : NG;            // rows per wave per iteration
  char* __restrict__ gxl_base = static_cast<char*>(p.gxl) + (size_t)ch0 * sizeof(T);
  const uint32_t ld_gxl = (uint32_t)(p.ld_gxl * sizeof(T));

  f32x2 a1[4], na1[4], a2[4], Pt[4], Qt[4], dbias[4];
  load_att(p.att, ch0, p.slope, kLog2e, a1, a2);
#pragma unroll
  for (int i = 0; i < 4; ++i) { na1[i] = -a1[i]; Pt[i] = splat(0.f); Qt[i] = splat(0.f); dbias[i] = splat(0.f); }

#pragma unroll 1
  for (int it = 0; it < p.rows_per_wave_iter; ++it) {
    // the block's four waves walk ADJACENT row batches in every iteration (like the forward's blocks) instead of each
    // wave its own run of consecutive batches: the rows in flight on a CU share neighbours (C2 pair -0.8 % bf16, -1.5 % fp32)
    const int64_t rbase = ((blk * (int64_t)p.rows_per_wave_iter + it) * 4 + L.wave) * RPW;
    if (rbase >= p.n_rows) break;              // wave-uniform
    const int64_t pos = rbase + (WPR ? 0 : L.grp);
    const bool row_ok = pos < p.n_rows;
    const int row = visit_row<WPR>(p.order, pos, row_ok);
    // the per-row loads / stores below address `matrix + row * ld + ch0`; with ch0 loop-invariant the compiler hoists one
    // 64-bit `matrix + ch0` pointer PER MATRIX out of this loop (9 VGPR pairs) and spills some of them.  An opaque copy of
    // ch0 per iteration keeps those sums inside it: scalar row base + 32-bit lane offset, no registers held across rows.
    int ch0r = L.ch0;
    asm volatile("" : "+v"(ch0r));
    const int ch0 = ch0r;                      // (shadows the function-scope ch0 for the rest of this iteration)

    f32x2 nxr[4], g[4], Sg[4];
    float D = 0.f, lse = 0.f, Sde = 0.f;
    int64_t beg = 0, end = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { nxr[i] = splat(0.f); g[i] = splat(0.f); Sg[i] = splat(0.f); }
    if (row_ok) {
      beg = p.indptr[row]; end = p.indptr[row + 1];
      f32x2 gy[4], pr[4], xr[4];
      load_pairs(static_cast<const T*>(p.xr) + row * p.ld_xr + ch0, xr);
#pragma unroll
      for (int i = 0; i < 4; ++i) nxr[i] = neg_canon_zero(xr[i]);
      load_pairs(static_cast<const T*>(p.gout) + row * p.ld_go + ch0, gy);
      load_pairs(static_cast<const T*>(p.pre) + row * p.ld_pre + ch0, pr);
      f32x2 d2 = splat(0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g[i] = p.apply_gelu ? f32x2{gy[i].x * gelu_erf_grad(pr[i].x), gy[i].y * gelu_erf_grad(pr[i].y)} : gy[i];
        f32x2 b = splat(0.f);
        if (p.bias) b = f32x2{p.bias[ch0 + 2 * i], p.bias[ch0 + 2 * i + 1]};
        d2 = pk_fma(g[i], pr[i] - b, d2);
      }
      D = d2.x + d2.y;
      lse = p.lse[(int64_t)row * H + h];
    }
    D = lane_block_sum<LPH>(D);
    const bool writer = row_ok && L.lane_on && (!WPR || L.grp == 0);
    if (writer) {
      if (p.zero_rows) {             // zero row of another matrix indexed by THESE rows (the pair form of the backward)
        f32x2 z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = splat(0.f);
        store_pairs(static_cast<T*>(p.zero_rows) + row * p.ld_zero + ch0, z);
      }
      store_pairs(static_cast<T*>(p.gpre) + row * p.ld_gp + ch0, g);
      if (head_leader)      // (lse, D) side by side: the source pass fetches both with one 8-byte load per (edge, head)
        *reinterpret_cast<float2*>(p.dsum + ((int64_t)row * H + h) * 2) = float2{lse, D};
#pragma unroll
      for (int i = 0; i < 4; ++i) dbias[i] = dbias[i] + g[i];
    }

    const int hbit = 1 << h;
    auto body = [&](auto meta_c, const bool (&valid)[U], const int (&nbr)[U], const int (&ed)[U]) {
      constexpr int META = decltype(meta_c)::value;
      if (!valid[0]) return;
      Raw8<T> raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u].load(row_ptr(xl, nbr[u], ld_xl));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f32x2 v[4], nt[4];
        raw[u].get(v);
        f32x2 da2 = splat(0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) { nt[i] = nxr[i] - v[i]; da2 = pk_fma(g[i], v[i], da2); }
        const float pl = lane_block_sum<LPH>(logit_partial(nt, na1, a2));
        float da = lane_block_sum<LPH>(da2.x + da2.y);
        const float a = valid[u] ? fast_exp2(pl - lse) : 0.f;
        float a_eff = a;
        if constexpr (META != kMetaNone) {
          const bool keep = META == kMetaBits ? (ed[u] & hbit) != 0
                                              : dropout_keep((uint32_t)ed[u], H, h, seed_lo, seed_hi, p.drop_thr);
          da = keep ? da * p.drop_scale : 0.f;
          a_eff = keep ? a * p.drop_scale : 0.f;
        }
        const float de = a * (da - D);
        Sde += de;
        const float nde = -de;
        const f32x2 nde2 = splat(nde);
        f32x2 gx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 sg = f32x2{sign_mul(nde, nt[i].x), sign_mul(nde, nt[i].y)};    // de * sgn(t)
          Pt[i] = pk_fma(nde2, nt[i], Pt[i]);                                        // de * t
          Qt[i] = pk_fma(-sg, nt[i], Qt[i]);                                         // de * |t|
          Sg[i] = Sg[i] + sg;
          if constexpr (DIRECT)      // the source pass's formula with this edge as the only term
            gx[i] = pk_fma(splat(a_eff), g[i], (a1[i] * de + a2[i] * sg) * kLn2);
        }
        if constexpr (DIRECT) {
          if (valid[u] && L.lane_on)
            store_pairs(reinterpret_cast<T*>(gxl_base + (uint64_t)(uint32_t)nbr[u] * ld_gxl), gx);
        }
      }
    };
    if (dropout && p.bits)
      walk_row<GS, WPR, kMetaBits, U>(p.col, p.bits, beg, end, L.lane, L.grp, L.gl, body);
    else if (dropout)
      walk_row<GS, WPR, kMetaEid, U>(p.col, p.eid, beg, end, L.lane, L.grp, L.gl, body);
    else
      walk_row<GS, WPR, kMetaNone, U>(p.col, nullptr, beg, end, L.lane, L.grp, L.gl, body);

    if constexpr (WPR) {
#pragma unroll
      for (int off = GS; off < 64; off <<= 1) {
        Sde += __shfl_xor(Sde, off, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Sg[i].x += __shfl_xor(Sg[i].x, off, 64);
          Sg[i].y += __shfl_xor(Sg[i].y, off, 64);
        }
      }
    }
    if (writer) {
      // a1 = att*c1*log2e, a2 = att*c2*log2e  ->  natural units via ln2
      f32x2 dxr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) dxr[i] = (a1[i] * Sde + a2[i] * Sg[i]) * kLn2;
      store_pairs(static_cast<T*>(p.gxr) + row * p.ld_gxr + ch0, dxr);
    }
  }

  // grad_att = c1 * Pt + c2 * Qt ; block partials of grad_att / grad_bias -> slab[blk]
  const float c1 = 0.5f * (1.f + p.slope), c2 = 0.5f * (1.f - p.slope);
  float datt[8], db[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    datt[2 * i] = c1 * Pt[i].x + c2 * Qt[i].x; datt[2 * i + 1] = c1 * Pt[i].y + c2 * Qt[i].y;
    db[2 * i] = dbias[i].x; db[2 * i + 1] = dbias[i].y;
  }
#pragma unroll
  for (int off = GS; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      datt[k] += __shfl_xor(datt[k], off, 64);
      db[k] += __shfl_xor(db[k], off, 64);
    }
  }
  if (L.grp == 0 && L.lane_on) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[L.wave][0][ch0 + k] = datt[k]; red[L.wave][1][ch0 + k] = db[k]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * HC; i += 256) {
    const int which = i / HC, c = i % HC;
    p.slab[blk * (2 * HC) + i] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

template <typename T, int H, int LPH, bool WPR, bool DIRECT>
__global__ __launch_bounds__(256, DIRECT ? 2 : SEGGER_BWD_DST_WAVES) void gatv2_bwd_dst_kernel(GatParams p) {
  gatv2_bwd_dst_body<T, H, LPH, WPR, DIRECT>(p, blockIdx.x);
}

// ============================================================================
// Backward, source side:  grad_xl   (rows = sources, col = destinations)
//   grad_xl[i] = sum_j keep/(1-p) * a_ij * g[j]  +  att * (c1 * sum_j de_ij + c2 * sum_j de_ij * sgn(t_ij))
// ============================================================================
template <typename T, int H, int LPH, bool WPR>
__device__ __forceinline__ void gatv2_bwd_src_body(SEGGER_BODY_PARAM p, int64_t bid) {
  using G = Geo<H, LPH>;
  constexpr int GS = G::GS, NG = G::NG, U = SEGGER_SRC_UNROLL < GS ? SEGGER_SRC_UNROLL : GS;
  const int64_t blk = xcd_remap(bid, p.nblocks_padded, p.nblocks);
  if (blk < 0) return;
  const LaneGeo L = lane_geo<G>();
  const int h = L.h, ch0 = L.ch0;
  const int64_t pos = WPR ? (blk * 4 + L.wave) : ((blk * 4 + L.wave) * NG + L.grp);
  const bool row_ok = pos < p.n_rows;
  const int row = visit_row<WPR>(p.order, pos, row_ok);
  const char* __restrict__ xr_base = static_cast<const char*>(p.xr) + (size_t)ch0 * sizeof(T);
  const char* __restrict__ g_base = static_cast<const char*>(p.gpre) + (size_t)ch0 * sizeof(T);
  const uint32_t ld_xr = (uint32_t)(p.ld_xr * sizeof(T)), ld_gp = (uint32_t)(p.ld_gp * sizeof(T));
  const bool dropout = p.drop_thr != 0;
  uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
  if (dropout && p.seed_dev) {
    const uint64_t mixed = splitmix64(p.seed_raw + *p.seed_dev);
    seed_lo = (uint32_t)mixed; seed_hi = (uint32_t)(mixed >> 32);
  }

  f32x2 a1[4], na1[4], a2[4], nv[4], acc[4], Sg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { nv[i] = splat(0.f); acc[i] = splat(0.f); Sg[i] = splat(0.f); }
  float Sde = 0.f;
  int64_t beg = 0, end = 0;
  if (row_ok) { beg = p.indptr[row]; end = p.indptr[row + 1]; }
  // tx-belongs-bd by source: most transcripts have no out-edge; such waves only write zeros
  if (__all(beg == end)) {
    if (row_ok && L.lane_on && (!WPR || L.grp == 0)) {
      store_pairs(static_cast<T*>(p.gxl) + row * p.ld_gxl + ch0, acc);
      if (p.zero_rows) store_pairs(static_cast<T*>(p.zero_rows) + row * p.ld_zero + ch0, acc);     // (acc is zero here)
    }
    return;
  }
  load_att(p.att, ch0, p.slope, kLog2e, a1, a2);
#pragma unroll
  for (int i = 0; i < 4; ++i) na1[i] = -a1[i];
  if (row_ok) {
    f32x2 v[4];
    load_pairs(static_cast<const T*>(p.xl) + row * p.ld_xl + ch0, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) nv[i] = neg_canon_zero(v[i]);
  }

  const int hbit = 1 << h;
  auto body = [&](auto meta_c, const bool (&valid)[U], const int (&nbr)[U], const int (&ed)[U]) {
    constexpr int META = decltype(meta_c)::value;
    if (!valid[0]) return;
    Raw8<T> rxr[U], rg[U];
    float lse[U], D[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rxr[u].load(row_ptr(xr_base, nbr[u], ld_xr));
      rg[u].load(row_ptr(g_base, nbr[u], ld_gp));
      const float2 ld = *reinterpret_cast<const float2*>(p.dsum + ((int64_t)nbr[u] * H + h) * 2);
      lse[u] = ld.x; D[u] = ld.y;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x2 xr[4], g[4], nt[4];
      rxr[u].get(xr);
      rg[u].get(g);
      f32x2 nda2 = splat(0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) { nt[i] = nv[i] - xr[i]; nda2 = pk_fma(g[i], nv[i], nda2); }
      const float pl = lane_block_sum<LPH>(logit_partial(nt, na1, a2));
      float da = -lane_block_sum<LPH>(nda2.x + nda2.y);
      const float a = valid[u] ? fast_exp2(pl - lse[u]) : 0.f;
      float a_eff = a;
      if constexpr (META != kMetaNone) {
        const bool keep = META == kMetaBits ? (ed[u] & hbit) != 0
                                            : dropout_keep((uint32_t)ed[u], H, h, seed_lo, seed_hi, p.drop_thr);
        da = keep ? da * p.drop_scale : 0.f;
        a_eff = keep ? a * p.drop_scale : 0.f;
      }
      const float de = a * (da - D[u]);
      Sde += de;
      const f32x2 ae2 = splat(a_eff);
      const float nde = -de;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = pk_fma(ae2, g[i], acc[i]);
        Sg[i] = Sg[i] + f32x2{sign_mul(nde, nt[i].x), sign_mul(nde, nt[i].y)};
      }
    }
  };
  if (dropout && p.bits)
    walk_row<GS, WPR, kMetaBits, U>(p.col, p.bits, beg, end, L.lane, L.grp, L.gl, body);
  else if (dropout)
    walk_row<GS, WPR, kMetaEid, U>(p.col, p.eid, beg, end, L.lane, L.grp, L.gl, body);
  else
    walk_row<GS, WPR, kMetaNone, U>(p.col, nullptr, beg, end, L.lane, L.grp, L.gl, body);

#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = acc[i] + (a1[i] * Sde + a2[i] * Sg[i]) * kLn2;
  if constexpr (WPR) {
#pragma unroll
    for (int off = GS; off < 64; off <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i].x += __shfl_xor(acc[i].x, off, 64);
        acc[i].y += __shfl_xor(acc[i].y, off, 64);
      }
    }
  }
  if (row_ok && L.lane_on && (!WPR || L.grp == 0)) {
    store_pairs(static_cast<T*>(p.gxl) + row * p.ld_gxl + ch0, acc);
    if (p.zero_rows) {                                     // zero row of another matrix indexed by the same sources
      f32x2 z[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = splat(0.f);
      store_pairs(static_cast<T*>(p.zero_rows) + row * p.ld_zero + ch0, z);
    }
  }
}

template <typename T, int H, int LPH, bool WPR>
__global__ __launch_bounds__(256, SEGGER_BWD_SRC_WAVES) void gatv2_bwd_src_kernel(GatParams p) {
  gatv2_bwd_src_body<T, H, LPH, WPR>(p, blockIdx.x);
}

// The source pass of one edge type (`a`: tx-neighbors-tx, group-per-row) and the ONE-PASS destination pass of another
// (`b`: tx-belongs-bd, wave-per-row, DIRECT) in one launch: b's few hundred latency-bound blocks go first and run beside
// a's.  Built for the source pass's 3 waves per SIMD with the one-pass body at 2 rows in flight (it spills ~24 registers
// there, which a 13 us kernel does not notice; at the one-pass form's own 2 waves per SIMD the merged launch measured
// +0.5 % on the captured 1M-edge step, as built -1.6 %).  Used for small batches only (SEGGER_BWD_PAIR_MAX_ROWS).
template <typename T, int H, int LPH>
__global__ __launch_bounds__(256, 3) void gatv2_bwd_src_dst_pair_kernel(GatParams a, GatParams b) {
  if ((int64_t)blockIdx.x < b.nblocks_padded) gatv2_bwd_dst_body<T, H, LPH, true, true, 2>(b, blockIdx.x);
  else gatv2_bwd_src_body<T, H, LPH, false>(a, (int64_t)blockIdx.x - b.nblocks_padded);
}

}  // namespace segger
