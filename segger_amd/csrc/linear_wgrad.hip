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
//
// DX variant (segger_linear_wgrad_dx): the DATA gradient dX[n, K] = dY[n, M] * W[M, K] of the same projection from the
// same LDS stage, so that dY -- the widest matrix of a layer's backward, [n, 3*HC] for the stacked projections -- is
// read from HBM once instead of twice (once by the dX GEMM, once here).  The staged dY rows are read a second time,
// row-major (ds_read_b128), as the B operand of v_mfma_f32_16x16x32 against W^T fragments that stay in registers for
// the whole kernel (wave w owns output columns 16*CT*w .. of dX: M/32 x CT fragments of 4 VGPRs); the 16 x K result of
// a stage goes through a double-buffered LDS tile so that it leaves as full rows, one 8- or 16-byte store per lane,
// issued one stage later (after the ring's barrier).  The stores share vmcnt with the LDS-DMA loads: the counted
// waits below include them.
#include <cstdlib>
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
  static __device__ __forceinline__ f32x4 run16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
  static __device__ __forceinline__ f32x4 run16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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
  const float* pn;           // GEN: normalised coordinate per row (x is not read: its rows are sinusoid features of pn)
  float log_max_period;
  const void* wt;            // DX: W^T [K, M] row-major in the activation dtype
  void* dx; int64_t ld_dx;   // DX: dX [n_rows, K]
  const void* gate; int64_t ld_gate;   // DX, optional: dX[row, c] *= gelu'(gate[row, c]) (the GELU the projection's input went through)
};

// One LDS-DMA wave-instruction: every lane fetches 16 bytes at rsrc + voffset + soffset (zeros when that is past
// the resource's range) and the wave's 1 KiB lands at LDS byte address lds_dst + 16 * lane.  Written in asm so
// that hipcc does not count it: it would otherwise drain vmcnt to 0 in front of every LDS read (it cannot tell
// which LDS bytes an asynchronous load will write), which is the pipelining this kernel lives on.  M0 carries the
// destination; it is saved and restored inside the statement (cdna_hip_programming.md 5.7).
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, uint32_t lds_dst, int voffset, int soffset) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
               "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, int64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return i32x4{(int32_t)(uint32_t)a, (int32_t)((a >> 32) & 0xffffu), (int32_t)bytes, 0x00020000};
}
constexpr int kNBuf = 4;                   // LDS ring: stages s .. s+2 in flight while stage s is consumed
constexpr int kAhead = kNBuf - 1;
constexpr int kOutOfRange = 0x40000000;    // byte offset past every slab (host checks slabs < 1 GiB): reads as zeros

// GEN: the X operand is not staged but generated from one float per row (the positional embedder's sinusoid features,
// X[row][f] = cos / sin(pn[row] * w_f)): the stage image is the dY rows followed by the 16 coordinates
template <int M, int K, int NW, bool GEN = false> struct WgGeo {
  static constexpr int SY = M * 2 + 64, SX = K * 2 + 64;          // LDS row strides in bytes
  static constexpr int IMG = GEN ? kStageRows * SY + kStageRows * 4
                                 : kStageRows * (SY + SX);         // one stage: dY rows, then X rows
  static constexpr int P = ((IMG + 1023) / 1024 + NW - 1) / NW;   // 1 KiB DMA chunks per wave per stage
  static constexpr int BUF = P * NW * 1024;                       // ring slot (chunks past IMG receive zeros)
  static constexpr int LDS = kNBuf * BUF;
  static constexpr int SO = K * 2 + 16;                           // DX: row stride of the dX stage tile
  static constexpr int OUT = kStageRows * SO;                     // DX: one dX stage tile (two of them behind the ring)
  // DX with a gate (dX *= gelu'(gate)): 16 rows of the gate matrix [n, K] ride behind the X rows of the stage image when
  // the ring slot has the room anyway (chunks past IMG are fetched -- as zeros -- regardless: no extra DMA instruction)
  static constexpr int IMG_G = IMG + kStageRows * SX;
  static constexpr bool GATE_OK = !GEN && IMG % 1024 == 0 && IMG_G <= BUF;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {   // lgkmcnt / expcnt untouched
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// The kernel body is a function (two of them can share a launch: wgrad_pair_kernel) and takes the parameter block by
// value (gatv2_kernels.h: by const reference a body came out of the register allocator slower); `lds` = the workgroup's
// shared array of at least wgrad_lds_bytes<...>() bytes, 1 KiB aligned; `bid` = the slab this workgroup owns.
template <int M, int K, int NW, bool GEN, bool DX>
constexpr int wgrad_lds_bytes() { return WgGeo<M, K, NW, GEN>::LDS + (DX ? 2 * WgGeo<M, K, NW, GEN>::OUT : 0); }

template <typename T, int M, int K, int NW, bool GEN = false, bool DX = false>
__device__ __forceinline__ void wgrad_body(const WgradParams p, const int64_t bid, unsigned char* lds) {
  using G = WgGeo<M, K, NW, GEN>;
  static_assert(!(GEN && DX), "the generated-operand form has no data gradient");
  // wave grid over (M tiles, K tiles); GEN: every wave its own K tiles, so no generated fragment is computed twice
  constexpr int WM = GEN ? 1 : (NW == 8 ? 4 : 2), WK = GEN ? NW : 2;
  static_assert(M * K / (64 * NW) <= 192, "accumulator does not fit the register file");
  constexpr int TM = M / 32, TK = K / 32;
  static_assert(TM % WM == 0 && TK % WK == 0, "tile grid does not split over the waves");
  constexpr int MT = TM / WM, KT = TK / WK;            // 32x32 tiles per wave
  constexpr int SY = G::SY, SX = G::SX, P = G::P, BUF = G::BUF;
  static_assert((kStageRows * SY) % 1024 == 0, "the dY / X boundary must fall on a DMA chunk boundary");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wk = wave / WM;
  const T* __restrict__ dy = static_cast<const T*>(p.dy);
  const T* __restrict__ x = static_cast<const T*>(p.x);

  // ---- staging by LDS-DMA through buffer resources based at this workgroup's first row: the source of a piece is
  //      an SGPR base + a per-lane 32-bit offset fixed for the whole kernel + a scalar stage offset; rows past the
  //      end of the matrix (the last, partial stage, and the stages the ring runs ahead of the slab) read as zeros
  const int64_t s_beg = bid * p.stages_per_block;
  int64_t s_end = s_beg + p.stages_per_block;
  if (s_end > p.n_stages) s_end = p.n_stages;
  const int n_local = (int)(s_end > s_beg ? s_end - s_beg : 0);
  const int64_t row_beg = s_beg * kStageRows;
  int64_t span_rows = p.n_rows > row_beg ? p.n_rows - row_beg : 0;             // valid rows from row_beg on ...
  if (span_rows > (int64_t)n_local * kStageRows) span_rows = (int64_t)n_local * kStageRows;   // ... inside this slab
  const i32x4 ry = make_rsrc(dy + row_beg * p.ld_dy, span_rows * p.ld_dy * 2);
  const i32x4 rx = GEN ? make_rsrc(p.pn + row_beg, span_rows * 4) : make_rsrc(x + row_beg * p.ld_x, span_rows * p.ld_x * 2);
  const int stage_bytes_y = kStageRows * (int)p.ld_dy * 2;
  const int stage_bytes_x = GEN ? kStageRows * 4 : kStageRows * (int)p.ld_x * 2;
  const bool gated = DX && G::GATE_OK && p.gate != nullptr;                     // (uniform over the launch)
  const i32x4 rg = gated ? make_rsrc(static_cast<const T*>(p.gate) + row_beg * p.ld_gate, span_rows * p.ld_gate * 2) : rx;
  const int stage_bytes_g = gated ? kStageRows * (int)p.ld_gate * 2 : 0;
  int voff[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int o = (wave + j * NW) * 1024 + lane * 16;  // byte offset of this lane's piece in the stage image
    if (o < kStageRows * SY) {
      const int row = o / SY, w = o % SY;
      voff[j] = row * (int)p.ld_dy * 2 + (w < M * 2 ? w : 0);      // a row's 64 pad bytes re-read its first bytes
    } else if (o < G::IMG) {
      const int oo = o - kStageRows * SY;
      if (GEN) {
        voff[j] = oo;                                      // the stage's 16 coordinates, contiguous
      } else {
        const int row = oo / SX, w = oo % SX;
        voff[j] = row * (int)p.ld_x * 2 + (w < K * 2 ? w : 0);
      }
    } else if (gated && o < G::IMG_G) {
      const int oo = o - G::IMG;
      const int row = oo / SX, w = oo % SX;
      voff[j] = row * (int)p.ld_gate * 2 + (w < K * 2 ? w : 0);
    } else {
      voff[j] = kOutOfRange;
    }
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue = [&](int local_stage) {                  // always P loads per wave: the vmcnt arithmetic below relies on it
    const uint32_t dst = lds_base + (uint32_t)((local_stage % kNBuf) * BUF + wave * 1024);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int co = (wave + j * NW) * 1024;                           // wave-uniform: which matrix this chunk belongs to
      const bool is_y = co < kStageRows * SY, is_g = DX && G::GATE_OK && co >= G::IMG;
      if (is_g) {                                                      // (compile-time false without a gate region)
        lds_dma16(rg, dst + j * NW * 1024, voff[j], local_stage * stage_bytes_g);
      } else {
        lds_dma16(is_y ? ry : rx, dst + j * NW * 1024, voff[j], local_stage * (is_y ? stage_bytes_y : stage_bytes_x));
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
  // GEN: this lane's feature columns f = 32 * (wk*KT + b) + (lane & 31): frequency in revolutions per unit coordinate,
  // cos for the first half of the columns, sin for the second (wave-uniform: a wave's K tiles lie in one half)
  float omega[KT];
  const bool gen_sin = GEN && (32 * wk * KT >= K / 2);
  if (GEN) {
    static_assert(!GEN || (K / 2) % (32 * KT) == 0, "a wave's feature tiles must not straddle the cos / sin halves");
#pragma unroll
    for (int b = 0; b < KT; ++b) {
      const int f = (32 * (wk * KT + b) + (lane & 31)) % (K / 2);
      omega[b] = expf(-p.log_max_period * (float)f / (float)(K / 2)) * 0.15915494309189535f;
    }
  }

  // ---- DX: this wave's W^T fragments (A operand of the 16x16x32 MFMA: lane (i, q) holds W^T[c0 + i][32 ks + 8 q ..+8)),
  //      resident for the whole kernel, and the addresses of its pieces of the dX stage tile
  constexpr int CT = DX ? (K / 16) / NW : 1;                // 16-column tiles of dX per wave
  constexpr int KS = DX ? M / 32 : 1;                       // k-steps of the dX product
  static_assert(!DX || ((K / 16) % NW == 0 && CT * KS * 4 <= 64), "dX tiles do not split over the waves");
  constexpr int SO = G::SO;
  constexpr int PB = kStageRows * K * 2 / (NW * 64);        // bytes of a dX stage tile per thread (8 or 16)
  static_assert(!DX || PB == 8 || PB == 16, "dX store width");
  u32x4 wf[CT][KS];
  unsigned char* lds_out = lds + G::LDS;
  const int dq = lane >> 4, dj = lane & 15;
  T* __restrict__ dxp = static_cast<T*>(p.dx);
  if constexpr (DX) {
    const T* wt = static_cast<const T*>(p.wt);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wf[ct][ks] = *reinterpret_cast<const u32x4*>(wt + (int64_t)(16 * (wave * CT + ct) + dj) * M + 32 * ks + 8 * dq);
  }
  // stage t's dX tile (LDS, written by all waves before the last barrier) -> global memory, full rows
  auto store_dx = [&](int t, bool check_rows) {
    const int o = tid * PB, row = o / (K * 2), colb = o % (K * 2);
    const unsigned char* src = lds_out + (t & 1) * G::OUT + row * SO + colb;
    const int64_t grow = row_beg + (int64_t)t * kStageRows + row;
    unsigned char* dst = reinterpret_cast<unsigned char*>(dxp + grow * p.ld_dx) + colb;
    if (PB == 8) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(src);
      if (!check_rows || grow < p.n_rows) *reinterpret_cast<u32x2*>(dst) = v;
    } else {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src);
      if (!check_rows || grow < p.n_rows) *reinterpret_cast<u32x4*>(dst) = v;
    }
  };

  for (int d = 0; d < kAhead; ++d) issue(d);
  for (int s = 0; s < n_local; ++s) {
    // this wave's pieces of stage s have landed once at most (kAhead - 1) * P of its loads are outstanding;
    // the barrier then makes every wave's pieces visible and retires all reads of stage s - 1, whose ring slot
    // the next issue overwrites
    if constexpr (!DX) {
      wait_vmcnt<(kAhead - 1) * P>();
      __syncthreads();
    } else {
      // the dX stores of earlier iterations sit between the loads in vmcnt's (in-order) queue: per wave the stream
      // is L0 L1 L2 | L3 | L4 S0 | L5 S1 | ..., so behind the loads of stage s there are 2 P loads and min(s - 1, 3)
      // stores (every in-loop store is unconditional: one instruction per wave and iteration, see store_dx)
      if (s < 2) wait_vmcnt<(kAhead - 1) * P>();
      else if (s == 2) wait_vmcnt<(kAhead - 1) * P + 1>();
      else if (s == 3) wait_vmcnt<(kAhead - 1) * P + 2>();
      else wait_vmcnt<(kAhead - 1) * P + 3>();
      // a bare barrier (+ the LDS writes of the previous iteration's dX tile drained): __syncthreads() would add a
      // release fence, i.e. vmcnt(0) -- the loads in flight are the pipelining this kernel lives on
      __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
    }
    issue(s + kAhead);
    if constexpr (DX) {
      if (s > 0) store_dx(s - 1, false);
    }
    const unsigned char* base = lds + (s % kNBuf) * BUF;
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
    if constexpr (GEN) {
      // B fragment of tile b: lane (n = lane & 31, h = lane >> 5) supplies X[row 8h + i][column n], i = 0..7
      const float* pr = reinterpret_cast<const float*>(base + kStageRows * SY) + 8 * (lane >> 5);
      const f32x4 pa = *reinterpret_cast<const f32x4*>(pr), pb = *reinterpret_cast<const f32x4*>(pr + 4);
      const float pv[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
      for (int b = 0; b < KT; ++b) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float rev = pv[i] * omega[b];
          v[i] = gen_sin ? __builtin_amdgcn_sinf(rev) : __builtin_amdgcn_cosf(rev);
        }
        const u32x4 fb = u32x4{Vec8<T>::pack(v[0], v[1]), Vec8<T>::pack(v[2], v[3]), Vec8<T>::pack(v[4], v[5]),
                               Vec8<T>::pack(v[6], v[7])};
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = WgMfma<T>::run(fa[a], fb, acc[a][b]);
      }
    } else {
#pragma unroll
      for (int b = 0; b < KT; ++b) {
        const u32x2 lo = lds_read_tr(base + off_x + b * 64);
        const u32x2 hi = lds_read_tr(base + off_x + b * 64 + 4 * SX);
        const u32x4 fb = u32x4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = WgMfma<T>::run(fa[a], fb, acc[a][b]);
      }
    }
    if constexpr (DX) {
      // dX^T tile = W^T[16 columns of dX, M] * dY^T[M, 16 rows]: B operand lane (j, q) = dY[row j][32 ks + 8 q ..+8),
      // a row-major 16-byte read of the staged rows; D lane (j, q) = dX[row j][c0 + 4 q + e], e = 0..3
      f32x4 dacc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) dacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 fy = *reinterpret_cast<const u32x4*>(base + dj * SY + (32 * ks + 8 * dq) * 2);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dacc[ct] = WgMfma<T>::run16(wf[ct][ks], fy, dacc[ct]);
      }
      unsigned char* ot = lds_out + (s & 1) * G::OUT + dj * SO;
      if (gated) {                                       // this lane's 4 columns of row dj of the staged gate rows
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const u32x2 gv = *reinterpret_cast<const u32x2*>(base + G::IMG + dj * SX + (16 * (wave * CT + ct) + 4 * dq) * 2);
          float z0, z1, z2, z3;
          Vec8<T>::unpack2(gv.x, z0, z1);
          Vec8<T>::unpack2(gv.y, z2, z3);
          dacc[ct] = f32x4{dacc[ct].x * gelu_erf_grad(z0), dacc[ct].y * gelu_erf_grad(z1),
                           dacc[ct].z * gelu_erf_grad(z2), dacc[ct].w * gelu_erf_grad(z3)};
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const u32x2 pk = u32x2{Vec8<T>::pack(dacc[ct].x, dacc[ct].y), Vec8<T>::pack(dacc[ct].z, dacc[ct].w)};
        *reinterpret_cast<u32x2*>(ot + (16 * (wave * CT + ct) + 4 * dq) * 2) = pk;
      }
    }
  }
  if constexpr (DX) {
    if (n_local > 0) {                                       // the last stage's tile (the only one that may be partial)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      store_dx(n_local - 1, true);
    }
  }
  // the ring ran kAhead stages past the slab (zero reads): let them land before the wave ends
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- partial results: acc tile (a, b) element e of lane l is dW[m][k] with
  //      m = 32*(wm*MT + a) + (e & 3) + 8*(e >> 2) + 4*(l >> 5),  k = 32*(wk*KT + b) + (l & 31) -----------------------
  float* out = p.partial + bid * (M * K + M);
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

template <typename T, int M, int K, int NW, bool GEN = false, bool DX = false>
__global__ __launch_bounds__(NW * 64) void wgrad_kernel(WgradParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[wgrad_lds_bytes<M, K, NW, GEN, DX>()];
  wgrad_body<T, M, K, NW, GEN, DX>(p, blockIdx.x, lds);
}

// Two projections' backward passes (same K, same wave count, both with or both without the data gradient) in ONE launch:
// the slabs of `b` (the boundary side of a hetero layer: a few hundred rows, one or two workgroups) are dispatched first
// and run beside the slabs of `a` instead of as a 7-15 us launch of their own (six such pairs in a captured 1M-edge step).
template <typename T, int MA, int MB, int K, int NW, bool DX>
__global__ __launch_bounds__(NW * 64) void wgrad_pair_kernel(WgradParams a, WgradParams b, int nb_b) {
  constexpr int LA = wgrad_lds_bytes<MA, K, NW, false, DX>(), LB = wgrad_lds_bytes<MB, K, NW, false, DX>();
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LA > LB ? LA : LB];
  if ((int)blockIdx.x < nb_b) wgrad_body<T, MB, K, NW, false, DX>(b, blockIdx.x, lds);
  else wgrad_body<T, MA, K, NW, false, DX>(a, (int64_t)blockIdx.x - nb_b, lds);
}

// ---- the positional embedder's whole backward from ONE read of its incoming gradient (segger_posmlp_bwd) -----------
// pe = SiLU(F(pn) W0^T + b0) W2^T + b2 per coordinate row (posmlp.hip); with g = d loss / d pe as rows [R = 2n, 64]:
//     dW2 = g^T h1,  db2 = sum g            h1 = SiLU(z1) recomputed from the stored pre-activation
//     dz1 = (g W2) * SiLU'(z1)              never leaves the chip
//     dW0 = dz1^T F(pn),  db0 = sum dz1     F regenerated from one float per row (the GEN form above)
// Round 2 ran this as three kernels (weight gradient 64x64, data gradient with the SiLU' epilogue, generated-operand
// weight gradient 64x256): g read twice, dz1 written and read back, h1 stored by the forward and read here.  One ring
// of 16-row stages {g rows, z1 rows, 16 coordinates} (LDS-DMA, 6 stages ahead, 8 slots so that a stage outlives the
// iteration after its own).  Iteration s PRODUCES, from stage s: the h1 tile (4 elements per thread) and the dz1 tile
// (v_mfma_f32_16x16x32 against W2^T fragments held in registers, times SiLU'), both into double-buffered LDS tiles in
// the staging row stride; it CONSUMES stage s - 1 behind the same barrier: dW2 += g^T h1 (both by transposing reads),
// dW0 += dz1^T F (A by transposing reads of the dz1 tile, B generated in the fragment layout), the bias sums from the A
// fragments.  One barrier per stage, no global stores inside the loop (so vmcnt counts the ring's loads only).
struct PosBwdParams {
  const void* g; int64_t ld_g;
  const void* z1;                    // [R, 64] contiguous
  const float* pn;                   // [R]
  const void* w2t;                   // W2^T [64 (column of z1), 64 (column of pe)] row-major
  int64_t n_rows, n_stages, stages_per_block;
  float* part2;                      // [grid][64*64 + 64]
  float* part0;                      // [grid][64*256 + 64]
  float log_max_period;
};
constexpr int kPbD = 64, kPbF = 256;
constexpr int kPbS = kPbD * 2 + 64;                       // LDS row stride of every 64-wide tile (== 192 mod 256)
constexpr int kPbTile = kStageRows * kPbS;                // 3072 = 3 DMA chunks
constexpr int kPbOffZ = kPbTile, kPbOffP = 2 * kPbTile;   // stage image: g rows | z1 rows | 16 coordinates
constexpr int kPbBuf = 8 * 1024;                       // one 1 KiB DMA chunk per wave (8 waves) per stage
// a stage is only 6 KB of payload: 3 stages ahead (as in wgrad_kernel) leave 2 x 18 KB in flight per CU, too little to
// cover the memory latency at this stage rate (measured 1.9 TB/s); 6 ahead, + the current stage + the one consume() reads
constexpr int kPbAhead = 6, kPbSlots = kPbAhead + 2;
constexpr int kPbWidth2 = kPbD * kPbD + kPbD, kPbWidth0 = kPbD * kPbF + kPbD;

__device__ __forceinline__ float sigmoid_fast(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}

// Two row sets (the transcripts' and the boundaries' positions: the embedder is called once per node type on large batches)
// share ONE launch and ONE sum of partials: blocks [0, n_first) work on `pa`, the rest on `pb`, whose partial rows follow
// pa's -- the embedder's four parameters receive one gradient each (no accumulation launches behind autograd's two nodes).
template <typename T>
__global__ __launch_bounds__(512, 2) void posmlp_bwd_kernel(PosBwdParams pa, PosBwdParams pb, int n_first) {
  const bool second = (int)blockIdx.x >= n_first;              // (uniform)
  const PosBwdParams& p = second ? pb : pa;
  const int bid = second ? (int)blockIdx.x - n_first : (int)blockIdx.x;
  // 8 waves: wave w stages DMA chunk w of every stage and owns feature tile w of dW0 (both 32-row tiles of dz1
  // columns); waves 0..3 also produce the dz1 tile (16 columns each) and own one tile of dW2, waves 4..7 produce the
  // h1 tile.  ~110 registers per lane -> 4 waves per SIMD: the loop is a dependent VALU / transcendental stream
  // (sin / cos, exp, rcp per element) that needs the other waves to fill its issue slots
  constexpr int D = kPbD, F = kPbF, S = kPbS, NW = 8, BUF = kPbBuf, NB = kPbSlots, TILE = kPbTile;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NB * BUF + 4 * TILE];
  unsigned char* tile_h = lds + NB * BUF;                   // [2][TILE]: h1 of stages s, s - 1
  unsigned char* tile_d = tile_h + 2 * TILE;                // [2][TILE]: dz1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* __restrict__ gp = static_cast<const T*>(p.g);
  const T* __restrict__ zp = static_cast<const T*>(p.z1);

  const int64_t s_beg = (int64_t)bid * p.stages_per_block;
  int64_t s_end = s_beg + p.stages_per_block;
  if (s_end > p.n_stages) s_end = p.n_stages;
  const int n_local = (int)(s_end > s_beg ? s_end - s_beg : 0);
  const int64_t row_beg = s_beg * kStageRows;
  int64_t span_rows = p.n_rows > row_beg ? p.n_rows - row_beg : 0;
  if (span_rows > (int64_t)n_local * kStageRows) span_rows = (int64_t)n_local * kStageRows;
  // this wave's chunk: 0..2 g rows, 3..5 z1 rows, 6 the coordinates, 7 padding (reads zeros)
  i32x4 rs; int stage_bytes, voff;
  {
    const int o = wave * 1024 + lane * 16;
    if (wave < 3) {
      rs = make_rsrc(gp + row_beg * p.ld_g, span_rows * p.ld_g * 2);
      stage_bytes = kStageRows * (int)p.ld_g * 2;
      const int row = o / S, w = o % S;
      voff = row * (int)p.ld_g * 2 + (w < D * 2 ? w : 0);
    } else if (wave < 6) {
      rs = make_rsrc(zp + row_beg * D, span_rows * D * 2);
      stage_bytes = kStageRows * D * 2;
      const int oo = o - kPbOffZ;
      const int row = oo / S, w = oo % S;
      voff = row * D * 2 + (w < D * 2 ? w : 0);
    } else {
      rs = make_rsrc(p.pn + row_beg, span_rows * 4);
      stage_bytes = kStageRows * 4;
      voff = (wave == 6 && lane < kStageRows / 4) ? lane * 16 : kOutOfRange;
    }
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue = [&](int local_stage) {                  // always one load per wave
    lds_dma16(rs, lds_base + (uint32_t)((local_stage % NB) * BUF + wave * 1024), voff, local_stage * stage_bytes);
  };

  // transposing-read address of this lane inside a 16-row tile of stride S (+ 64 bytes per 32-column tile)
  const int g4 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const int off_a = (8 * (g4 >> 1) + tq) * S + (16 * (g4 & 1) + 4 * tp) * 2;
  const int wm2 = wave & 1, wk2 = (wave >> 1) & 1;     // waves 0..3: their 32x32 tile of dW2 (pe columns x h1 columns)
  const int dq = lane >> 4, dj = lane & 15;            // waves 0..3: dz1 lane (row dj, columns 16 wave + 4 dq ..+4)
  const int r = lane & 31, h = lane >> 5;
  const bool lo_half = wave < 4;                        // (wave-uniform)

  f32x16 acc2, acc0[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc2[e] = 0.f; acc0[0][e] = 0.f; acc0[1][e] = 0.f; }
  float db2 = 0.f, db0[2] = {0.f, 0.f};
  // this wave's feature columns f = 32 wave + r: revolutions per unit coordinate; cos for f < 128, sin above
  const float omega = expf(-p.log_max_period * (float)((32 * wave + r) % (F / 2)) / (float)(F / 2)) * 0.15915494309189535f;
  u32x4 wf[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};  // W2^T rows 16 wave + dj, k-steps of 32 pe columns
  if (lo_half) {
    const T* wt = static_cast<const T*>(p.w2t);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[ks] = *reinterpret_cast<const u32x4*>(wt + (16 * wave + dj) * D + 32 * ks + 8 * dq);
  }
  // these are the kernel's only compiler-visible global loads: retire them here, or hipcc -- which cannot see the
  // asm DMA loads sharing the counter -- waits for them with vmcnt(0) at their first use INSIDE the loop, draining the
  // ring every iteration (seen in the ISA of the first version: 1.9 TB/s)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" : "+v"(wf[0]), "+v"(wf[1]));

  auto consume = [&](int t) {
    const unsigned char* pb = lds + (t % NB) * BUF;
    const unsigned char* th = tile_h + (t & 1) * TILE;
    const unsigned char* td = tile_d + (t & 1) * TILE;
    if (lo_half) {
      const u32x2 lo = lds_read_tr(pb + off_a + wm2 * 64), hi = lds_read_tr(pb + off_a + wm2 * 64 + 4 * S);
      const u32x4 fa = u32x4{lo.x, lo.y, hi.x, hi.y};
      if (wk2 == 0) {
        db2 = WgMfma<T>::sum2(fa.x, db2); db2 = WgMfma<T>::sum2(fa.y, db2);
        db2 = WgMfma<T>::sum2(fa.z, db2); db2 = WgMfma<T>::sum2(fa.w, db2);
      }
      const u32x2 lo2 = lds_read_tr(th + off_a + wk2 * 64), hi2 = lds_read_tr(th + off_a + wk2 * 64 + 4 * S);
      acc2 = WgMfma<T>::run(fa, u32x4{lo2.x, lo2.y, hi2.x, hi2.y}, acc2);
    }
    u32x4 fd[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const u32x2 lo = lds_read_tr(td + off_a + a * 64), hi = lds_read_tr(td + off_a + a * 64 + 4 * S);
      fd[a] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    if (wave == 7) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float d = db0[a];
        d = WgMfma<T>::sum2(fd[a].x, d); d = WgMfma<T>::sum2(fd[a].y, d);
        d = WgMfma<T>::sum2(fd[a].z, d); d = WgMfma<T>::sum2(fd[a].w, d);
        db0[a] = d;
      }
    }
    // B fragment of this wave's feature tile: lane (column r, half h) supplies F[row 8 h + i][f], i = 0..7
    const float* pr = reinterpret_cast<const float*>(pb + kPbOffP) + 8 * h;
    const f32x4 pa = *reinterpret_cast<const f32x4*>(pr), pc = *reinterpret_cast<const f32x4*>(pr + 4);
    const float pv[8] = {pa.x, pa.y, pa.z, pa.w, pc.x, pc.y, pc.z, pc.w};
    float v[8];
    if (lo_half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_cosf(pv[i] * omega);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_sinf(pv[i] * omega);
    }
    const u32x4 fb = u32x4{Vec8<T>::pack(v[0], v[1]), Vec8<T>::pack(v[2], v[3]), Vec8<T>::pack(v[4], v[5]),
                           Vec8<T>::pack(v[6], v[7])};
#pragma unroll
    for (int a = 0; a < 2; ++a) acc0[a] = WgMfma<T>::run(fd[a], fb, acc0[a]);
  };

  for (int d = 0; d < kPbAhead; ++d) issue(d);
  for (int s = 0; s < n_local; ++s) {
    wait_vmcnt<kPbAhead - 1>();
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): last iteration's tile writes and reads
    __builtin_amdgcn_s_barrier();
    issue(s + kPbAhead);                                   // the slot consume(s - 2) read last
    const unsigned char* base = lds + (s % NB) * BUF;
    if (!lo_half) {                                        // h1 tile: 4 elements per thread of waves 4..7
      const int t4 = tid - 256, row = t4 >> 4, c4 = (t4 & 15) * 4;
      const u32x2 zv = *reinterpret_cast<const u32x2*>(base + kPbOffZ + row * S + c4 * 2);
      float z0, z1, z2, z3;
      Vec8<T>::unpack2(zv.x, z0, z1);
      Vec8<T>::unpack2(zv.y, z2, z3);
      const u32x2 hv = u32x2{Vec8<T>::pack(z0 * sigmoid_fast(z0), z1 * sigmoid_fast(z1)),
                             Vec8<T>::pack(z2 * sigmoid_fast(z2), z3 * sigmoid_fast(z3))};
      *reinterpret_cast<u32x2*>(tile_h + (s & 1) * TILE + row * S + c4 * 2) = hv;
    } else {                                               // dz1 tile: 16 columns per wave of waves 0..3
      f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 fy = *reinterpret_cast<const u32x4*>(base + dj * S + (32 * ks + 8 * dq) * 2);
        dacc = WgMfma<T>::run16(wf[ks], fy, dacc);
      }
      const u32x2 zv = *reinterpret_cast<const u32x2*>(base + kPbOffZ + dj * S + (16 * wave + 4 * dq) * 2);
      float z[4];
      Vec8<T>::unpack2(zv.x, z[0], z[1]);
      Vec8<T>::unpack2(zv.y, z[2], z[3]);
      float o[4] = {dacc.x, dacc.y, dacc.z, dacc.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = sigmoid_fast(z[e]);
        o[e] *= sg * (1.0f + z[e] * (1.0f - sg));          // SiLU'(z)
      }
      *reinterpret_cast<u32x2*>(tile_d + (s & 1) * TILE + dj * S + (16 * wave + 4 * dq) * 2) =
          u32x2{Vec8<T>::pack(o[0], o[1]), Vec8<T>::pack(o[2], o[3])};
    }
    if (s > 0) consume(s - 1);
  }
  if (n_local > 0) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    consume(n_local - 1);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                      // the ring ran kPbAhead stages past the slab

  float* out2 = p.part2 + (int64_t)bid * kPbWidth2;
  float* out0 = p.part0 + (int64_t)bid * kPbWidth0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
    if (lo_half) out2[(32 * wm2 + m) * D + 32 * wk2 + r] = acc2[e];
#pragma unroll
    for (int a = 0; a < 2; ++a) out0[(32 * a + m) * F + 32 * wave + r] = acc0[a][e];
  }
  if (lo_half && wk2 == 0) {
    const float d = db2 + __shfl_xor(db2, 32, 64);
    if (h == 0) out2[D * D + 32 * wm2 + r] = d;
  }
  if (wave == 7) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float d = db0[a] + __shfl_xor(db0[a], 32, 64);
      if (h == 0) out0[D * F + 32 * a + r] = d;
    }
  }
}

// out[e] = sum_s partial[s][e] in two deterministic stages (a single pass would leave each of the few thousand
// threads a serial chain of up to 1024 dependent-latency loads): stage 1 sums every kRedGroups-th slab into
// part2[g][e] (grid = columns x groups), stage 2 adds the groups in order; grad_w = first M*K entries, grad_b the rest
constexpr int kRedGroups = 32;
__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float* __restrict__ partial, int64_t n_slabs, int64_t width,
                                                           float* __restrict__ part2) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= width) return;
  const int g = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t s = g;
  for (; s + 3 * kRedGroups < n_slabs; s += 4 * kRedGroups) {
    s0 += partial[s * width + e];                    s1 += partial[(s + kRedGroups) * width + e];
    s2 += partial[(s + 2 * kRedGroups) * width + e]; s3 += partial[(s + 3 * kRedGroups) * width + e];
  }
  for (; s < n_slabs; s += kRedGroups) s0 += partial[s * width + e];
  part2[(int64_t)g * width + e] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void wgrad_reduce2_kernel(const float* __restrict__ part2, int64_t width, int64_t mk,
                                                           float* __restrict__ grad_w, float* __restrict__ grad_b) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= width) return;
  // (the order of reduce_many_kernel over the same 32 groups, csrc/reduce.hip: a sum comes out bit-identical whether it
  // was launched here or queued and flushed with others)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int g = 0; g < kRedGroups; g += 4) {
    s0 += part2[(int64_t)g * width + e];       s1 += part2[(int64_t)(g + 1) * width + e];
    s2 += part2[(int64_t)(g + 2) * width + e]; s3 += part2[(int64_t)(g + 3) * width + e];
  }
  const float t = (s0 + s1) + (s2 + s3);
  if (e < mk) grad_w[e] = t;
  else if (grad_b) grad_b[e - mk] = t;
}

// few slabs (small batches): one pass, each thread sums its column over all slabs in slab order
constexpr int kSinglePassSlabs = 128;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int64_t n_slabs, int64_t width,
                                                          int64_t mk, float* __restrict__ grad_w, float* __restrict__ grad_b) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= width) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t s = 0;
  for (; s + 3 < n_slabs; s += 4) {
    s0 += partial[s * width + e];       s1 += partial[(s + 1) * width + e];
    s2 += partial[(s + 2) * width + e]; s3 += partial[(s + 3) * width + e];
  }
  for (; s < n_slabs; ++s) s0 += partial[s * width + e];
  const float t = (s0 + s1) + (s2 + s3);
  if (e < mk) grad_w[e] = t;
  else if (grad_b) grad_b[e - mk] = t;
}

constexpr int kNumCu = 256;
// 512 rows: below that a workgroup's partial [M*K] costs more than its GEMM (SEGGER_WGRAD_MIN_STAGES overrides: tools)
static const int kMinStagesPerBlock = [] {
  const char* e = getenv("SEGGER_WGRAD_MIN_STAGES");
  const int v = e ? atoi(e) : 0;
  return v >= 4 && v <= 1024 ? v : 32;
}();

bool shape_ok(int m, int k) {
  return (m == 384 || m == 192 || m == 128 || m == 64) && (k == 256 || k == 128 || k == 64);
}
int waves_for(int m) { return m % 128 == 0 ? 8 : 4; }
// workgroups per CU the register and LDS footprints allow (accumulator registers per lane = M*K / (64*NW);
// LDS = 4 ring slots of the stage image rounded up to whole 1 KiB chunks per wave)
int blocks_per_cu(int m, int k, bool dx = false) {
  const int nw = waves_for(m);
  int acc = m * k / (64 * nw);
  const int img = kStageRows * (m * 2 + 64 + k * 2 + 64);
  int lds = kNBuf * (((img + 1023) / 1024 + nw - 1) / nw) * nw * 1024;
  if (dx) {                                            // + the W^T fragments and the two dX stage tiles
    acc += (k / 16 / nw) * (m / 32) * 4;
    lds += 2 * kStageRows * (k * 2 + 16);
  }
  int by_regs = acc > 64 ? 1 : (acc > 32 ? 2 : 4);
  if (nw == 4 && by_regs < 4) by_regs *= 2;           // 4-wave workgroups: one wave per SIMD each
  const int by_lds = (160 * 1024) / lds;
  const int b = by_regs < by_lds ? by_regs : by_lds;
  return b < 1 ? 1 : b;
}
int64_t grid_for(int64_t n_rows, int m, int k, bool dx = false) {
  const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
  const int64_t cap = (int64_t)kNumCu * blocks_per_cu(m, k, dx);
  const int64_t want = (stages + kMinStagesPerBlock - 1) / kMinStagesPerBlock;
  return want < cap ? (want < 1 ? 1 : want) : cap;
}

template <typename T, int M, int K>
void launch_wgrad(const WgradParams& p, int64_t grid, hipStream_t stream) {
  constexpr int NW = M % 128 == 0 ? 8 : 4;
  hipLaunchKernelGGL((wgrad_kernel<T, M, K, NW>), dim3((unsigned)grid), dim3(NW * 64), 0, stream, p);
}

int reduce_partials(float* partial, int64_t grid, int m_out, int k_in, float* grad_w, float* grad_b, hipStream_t stream);

template <typename T, int M, int K>
void launch_wgrad_dx(const WgradParams& p, int64_t grid, hipStream_t stream) {
  constexpr int NW = M % 128 == 0 ? 8 : 4;
  hipLaunchKernelGGL((wgrad_kernel<T, M, K, NW, false, true>), dim3((unsigned)grid), dim3(NW * 64), 0, stream, p);
}

bool dx_shape_ok(int m, int k) { return k == 128 && (m == 384 || m == 192 || m == 128 || m == 64); }
bool dx_gate_ok(int m, int k) {          // WgGeo::GATE_OK of the instantiated shapes
  return k == 128 && ((m == 384 && WgGeo<384, 128, 8>::GATE_OK) || (m == 128 && WgGeo<128, 128, 8>::GATE_OK));
}

template <typename T>
int dispatch_wgrad_dx(const WgradParams& p, int m, int64_t grid, hipStream_t stream) {
  switch (m) {
    case 384: launch_wgrad_dx<T, 384, 128>(p, grid, stream); return SEGGER_OK;
    case 192: launch_wgrad_dx<T, 192, 128>(p, grid, stream); return SEGGER_OK;
    case 128: launch_wgrad_dx<T, 128, 128>(p, grid, stream); return SEGGER_OK;
    case 64:  launch_wgrad_dx<T, 64, 128>(p, grid, stream); return SEGGER_OK;
  }
  set_error("segger_linear_wgrad_dx: m_out=%d not supported", m);
  return SEGGER_EUNSUPPORTED;
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
  return shape_ok(m_out, k_in) && (dtype == SEGGER_BF16 || dtype == SEGGER_F16 || dtype == SEGGER_F32);
}

extern "C" size_t segger_linear_wgrad_workspace_bytes(int64_t n_rows, int32_t m_out, int32_t k_in) {
  if (n_rows <= 0 || !shape_ok(m_out, k_in)) return 16;
  const size_t b16 = (size_t)(grid_for(n_rows, m_out, k_in) + kRedGroups) * ((size_t)m_out * k_in + m_out) * sizeof(float);
  size_t b32 = wgrad_f32_workspace_bytes(n_rows, m_out, k_in);             // (the fp32 kernels' slab counts)
  if (wgrad_f32_split_shape_ok(m_out, k_in)) {
    const size_t bs = (size_t)(wgrad_f32_split_grid(n_rows, m_out, k_in) + kRedGroups) * ((size_t)m_out * k_in + m_out) * sizeof(float);
    b32 = bs > b32 ? bs : b32;
  }
  return b16 > b32 ? b16 : b32;
}

extern "C" int segger_linear_wgrad_dx_supported(int32_t m_out, int32_t k_in, int32_t dtype) {
  return dx_shape_ok(m_out, k_in) && (dtype == SEGGER_BF16 || dtype == SEGGER_F16);
}

extern "C" int segger_linear_wgrad_dx_gate_supported(int32_t m_out, int32_t k_in, int32_t dtype) {
  return segger_linear_wgrad_dx_supported(m_out, k_in, dtype) && dx_gate_ok(m_out, k_in);
}

extern "C" int segger_linear_wgrad_dx(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, const void* w_t,
                                      int64_t n_rows, int32_t m_out, int32_t k_in, int32_t dtype, float* grad_w,
                                      float* grad_b, void* dx, int64_t ld_dx, const void* gelu_gate, int64_t ld_gate,
                                      void* workspace, size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_rows >= 0 && m_out > 0 && k_in > 0, "segger_linear_wgrad_dx: bad sizes");
  SEGGER_REQUIRE(grad_w != nullptr, "segger_linear_wgrad_dx: grad_w is NULL");
  if (!segger_linear_wgrad_dx_supported(m_out, k_in, dtype)) {
    set_error("segger_linear_wgrad_dx: m_out=%d k_in=%d dtype=%d not supported (m_out in {64,128,192,384}, k_in 128, "
              "bf16/f16)", m_out, k_in, dtype);
    return SEGGER_EUNSUPPORTED;
  }
  if (n_rows == 0) {
    SEGGER_HIP(hipMemsetAsync(grad_w, 0, (size_t)m_out * k_in * sizeof(float), stream));
    if (grad_b) SEGGER_HIP(hipMemsetAsync(grad_b, 0, (size_t)m_out * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(dy && x && w_t && dx, "segger_linear_wgrad_dx: NULL pointer");
  SEGGER_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(w_t) && aligned16(dx),
                 "segger_linear_wgrad_dx: pointers must be 16-byte aligned");
  SEGGER_REQUIRE(ld_dy >= m_out && ld_x >= k_in && ld_dx >= k_in && (ld_dy * 2) % 16 == 0 && (ld_x * 2) % 16 == 0 &&
                     (ld_dx * 2) % 16 == 0, "segger_linear_wgrad_dx: bad leading dimension");
  if (gelu_gate) {
    SEGGER_REQUIRE(dx_gate_ok(m_out, k_in), "segger_linear_wgrad_dx: the gate form covers m_out in {128, 384} (k_in 128)");
    SEGGER_REQUIRE(aligned16(gelu_gate) && ld_gate >= k_in && (ld_gate * 2) % 16 == 0,
                   "segger_linear_wgrad_dx: gate rows must be 16-byte aligned");
  }
  const size_t need = segger_linear_wgrad_workspace_bytes(n_rows, m_out, k_in);     // (the dX form never uses more slabs)
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_linear_wgrad_dx: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t grid = grid_for(n_rows, m_out, k_in, true);
  const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
  {
    const int64_t span = ((stages + grid - 1) / grid) * kStageRows * (ld_dy > ld_x ? ld_dy : ld_x) * 2;
    SEGGER_REQUIRE(span < (int64_t)kOutOfRange, "segger_linear_wgrad_dx: a workgroup's row slab exceeds 1 GiB");
  }
  {
    const int64_t span_g = ((stages + grid - 1) / grid) * kStageRows * ld_gate * 2;
    SEGGER_REQUIRE(!gelu_gate || span_g < (int64_t)kOutOfRange, "segger_linear_wgrad_dx: a workgroup's row slab exceeds 1 GiB");
  }
  WgradParams p{dy, ld_dy, x, ld_x, n_rows, stages, (stages + grid - 1) / grid, static_cast<float*>(workspace), nullptr, 0.f,
                w_t, dx, ld_dx, gelu_gate, ld_gate};
  const int rc = dtype == SEGGER_BF16 ? dispatch_wgrad_dx<bf16_t>(p, m_out, grid, stream)
                                      : dispatch_wgrad_dx<f16_t>(p, m_out, grid, stream);
  if (rc != SEGGER_OK) return rc;
  SEGGER_LAUNCH_CHECK("wgrad_kernel (dX)");
  return reduce_partials(p.partial, grid, m_out, k_in, grad_w, grad_b, stream);
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
  const size_t need = segger_linear_wgrad_workspace_bytes(n_rows, m_out, k_in);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_linear_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  if (dtype == SEGGER_F32) {       // fp32 storage: exact-fp32 MFMA kernel (csrc/linear_f32.hip), same partial sums
    SEGGER_REQUIRE(ld_dy >= m_out && ld_x >= k_in, "segger_linear_wgrad: bad leading dimension");
    int64_t slabs = 0;
    const int rc32 = wgrad_f32_launch(dy, ld_dy, x, ld_x, n_rows, m_out, k_in, static_cast<float*>(workspace), &slabs, stream);
    if (rc32 != SEGGER_OK) return rc32;
    return reduce_partials(static_cast<float*>(workspace), slabs, m_out, k_in, grad_w, grad_b, stream);
  }
  SEGGER_REQUIRE(ld_dy >= m_out && ld_x >= k_in && (ld_dy * 2) % 16 == 0 && (ld_x * 2) % 16 == 0,
                 "segger_linear_wgrad: bad leading dimension");
  const int64_t grid = grid_for(n_rows, m_out, k_in);
  {
    // buffer resources address one workgroup's slab with 32-bit offsets
    const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
    const int64_t span = ((stages + grid - 1) / grid) * kStageRows * (ld_dy > ld_x ? ld_dy : ld_x) * 2;
    SEGGER_REQUIRE(span < (int64_t)kOutOfRange, "segger_linear_wgrad: a workgroup's row slab exceeds 1 GiB");
  }
  WgradParams p{dy, ld_dy, x, ld_x, n_rows, (n_rows + kStageRows - 1) / kStageRows, 0, static_cast<float*>(workspace)};
  p.stages_per_block = (p.n_stages + grid - 1) / grid;
  const int rc = dtype == SEGGER_BF16 ? dispatch_wgrad<bf16_t>(p, m_out, k_in, grid, stream)
                                      : dispatch_wgrad<f16_t>(p, m_out, k_in, grid, stream);
  if (rc != SEGGER_OK) return rc;
  SEGGER_LAUNCH_CHECK("wgrad_kernel");
  return reduce_partials(p.partial, grid, m_out, k_in, grad_w, grad_b, stream);
}

// ---- two backward passes in one launch -----------------------------------------------------------------------------
namespace segger {
namespace {
template <typename T>
bool launch_wgrad_pair(const WgradParams& a, int ma, int64_t grid_a, const WgradParams& b, int mb, int64_t grid_b, int k,
                       bool dx, hipStream_t stream) {
  const dim3 grid((unsigned)(grid_a + grid_b));
#define PAIR(MA, MB, KK, NWW, DXX)                                                                                     \
  if (ma == MA && mb == MB && k == KK && dx == DXX) {                                                                  \
    hipLaunchKernelGGL((wgrad_pair_kernel<T, MA, MB, KK, NWW, DXX>), grid, dim3(NWW * 64), 0, stream, a, b, (int)grid_b); \
    return true;                                                                                                       \
  }
  // the pairs of the default encoder: a hetero layer ([lin_l | lin_r | lin_l] of the transcripts + lin_r of the
  // boundaries; the first layer reads K = 256 and has no one-pass data gradient) and lin_last of both node types
  PAIR(384, 128, 128, 8, true) PAIR(384, 128, 256, 8, false) PAIR(64, 64, 128, 4, true)
  PAIR(384, 128, 128, 8, false) PAIR(64, 64, 128, 4, false)
#undef PAIR
  return false;
}
}  // namespace
}  // namespace segger

extern "C" int segger_linear_wgrad_f32_split(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t n_rows,
                                             int32_t m_out, int32_t k_in, float* grad_w, float* grad_b, void* workspace,
                                             size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_rows >= 0 && grad_w != nullptr, "segger_linear_wgrad_f32_split: bad sizes or NULL grad_w");
  if (!wgrad_f32_split_shape_ok(m_out, k_in)) {
    set_error("segger_linear_wgrad_f32_split: m_out=%d k_in=%d not supported", m_out, k_in);
    return SEGGER_EUNSUPPORTED;
  }
  if (n_rows == 0) {
    SEGGER_HIP(hipMemsetAsync(grad_w, 0, (size_t)m_out * k_in * sizeof(float), stream));
    if (grad_b) SEGGER_HIP(hipMemsetAsync(grad_b, 0, (size_t)m_out * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(dy && x && aligned16(dy) && aligned16(x) && ld_dy >= m_out && ld_x >= k_in && ld_dy % 4 == 0 && ld_x % 4 == 0,
                 "segger_linear_wgrad_f32_split: rows must be 16-byte aligned");
  const size_t need = segger_linear_wgrad_workspace_bytes(n_rows, m_out, k_in);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_linear_wgrad_f32_split: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  int64_t slabs = 0;
  const int rc = wgrad_f32_split_launch(dy, ld_dy, x, ld_x, n_rows, m_out, k_in, static_cast<float*>(workspace), &slabs, stream);
  if (rc != SEGGER_OK) return rc;
  return reduce_partials(static_cast<float*>(workspace), slabs, m_out, k_in, grad_w, grad_b, stream);
}

extern "C" int segger_linear_wgrad_pair(const segger_wgrad_args* a, const segger_wgrad_args* b, int32_t k_in, int32_t dtype,
                                        segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(a && b, "segger_linear_wgrad_pair: NULL args");
  const segger_wgrad_args* both[2] = {a, b};
  const bool dx = a->w_t != nullptr;
  auto separately = [&]() -> int {
    for (const segger_wgrad_args* q : both) {
      const int rc = q->w_t ? segger_linear_wgrad_dx(q->dy, q->ld_dy, q->x, q->ld_x, q->w_t, q->n_rows, q->m_out, k_in, dtype,
                                                     q->grad_w, q->grad_b, q->dx, q->ld_dx, nullptr, 0, q->workspace,
                                                     q->workspace_bytes, stream_)
                            : segger_linear_wgrad(q->dy, q->ld_dy, q->x, q->ld_x, q->n_rows, q->m_out, k_in, dtype, q->grad_w,
                                                  q->grad_b, q->workspace, q->workspace_bytes, stream_);
      if (rc != SEGGER_OK) return rc;
    }
    return SEGGER_OK;
  };
  if ((dtype != SEGGER_BF16 && dtype != SEGGER_F16) || (b->w_t != nullptr) != dx || a->n_rows <= 0 || b->n_rows <= 0)
    return separately();
  WgradParams p[2];
  int64_t grid[2];
  for (int i = 0; i < 2; ++i) {
    const segger_wgrad_args* q = both[i];
    const bool ok = dx ? segger_linear_wgrad_dx_supported(q->m_out, k_in, dtype) : segger_linear_wgrad_supported(q->m_out, k_in, dtype);
    if (!ok) return separately();
    SEGGER_REQUIRE(q->grad_w != nullptr, "segger_linear_wgrad_pair: grad_w is NULL");
    SEGGER_REQUIRE(q->dy && q->x && (!dx || q->dx), "segger_linear_wgrad_pair: NULL pointer");
    SEGGER_REQUIRE(aligned16(q->dy) && aligned16(q->x) && aligned16(q->w_t) && aligned16(q->dx),
                   "segger_linear_wgrad_pair: pointers must be 16-byte aligned");
    SEGGER_REQUIRE(q->ld_dy >= q->m_out && q->ld_x >= k_in && (q->ld_dy * 2) % 16 == 0 && (q->ld_x * 2) % 16 == 0 &&
                       (!dx || (q->ld_dx >= k_in && (q->ld_dx * 2) % 16 == 0)), "segger_linear_wgrad_pair: bad leading dimension");
    const size_t need = segger_linear_wgrad_workspace_bytes(q->n_rows, q->m_out, k_in);
    if (q->workspace == nullptr || q->workspace_bytes < need) {
      set_error("segger_linear_wgrad_pair: workspace %zu < %zu bytes", q->workspace_bytes, need);
      return SEGGER_EWORKSPACE;
    }
    grid[i] = grid_for(q->n_rows, q->m_out, k_in, dx);
    const int64_t stages = (q->n_rows + kStageRows - 1) / kStageRows;
    const int64_t span = ((stages + grid[i] - 1) / grid[i]) * kStageRows * (q->ld_dy > q->ld_x ? q->ld_dy : q->ld_x) * 2;
    SEGGER_REQUIRE(span < (int64_t)kOutOfRange, "segger_linear_wgrad_pair: a workgroup's row slab exceeds 1 GiB");
    p[i] = WgradParams{q->dy, q->ld_dy, q->x, q->ld_x, q->n_rows, stages, (stages + grid[i] - 1) / grid[i],
                       static_cast<float*>(q->workspace), nullptr, 0.f, q->w_t, q->dx, q->ld_dx, nullptr, 0};
  }
  const bool launched = dtype == SEGGER_BF16
      ? launch_wgrad_pair<bf16_t>(p[0], a->m_out, grid[0], p[1], b->m_out, grid[1], k_in, dx, stream)
      : launch_wgrad_pair<f16_t>(p[0], a->m_out, grid[0], p[1], b->m_out, grid[1], k_in, dx, stream);
  if (!launched) return separately();
  SEGGER_LAUNCH_CHECK("wgrad_pair_kernel");
  for (int i = 0; i < 2; ++i) {
    const int rc = reduce_partials(p[i].partial, grid[i], both[i]->m_out, k_in, both[i]->grad_w, both[i]->grad_b, stream);
    if (rc != SEGGER_OK) return rc;
  }
  return SEGGER_OK;
}

namespace segger {
namespace {
int reduce_partials(float* partial, int64_t grid, int m_out, int k_in, float* grad_w, float* grad_b, hipStream_t stream) {
  const int64_t width = (int64_t)m_out * k_in + m_out;
  if (defer_reduce(ReduceSeg{partial, grid, width, (int64_t)m_out * k_in, grad_w, grad_b, partial + grid * width}, stream))
    return SEGGER_OK;
  if (grid <= kSinglePassSlabs) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, stream, partial, grid,
                       width, (int64_t)m_out * k_in, grad_w, grad_b);
    SEGGER_LAUNCH_CHECK("wgrad_reduce_kernel");
    return SEGGER_OK;
  }
  float* part2 = partial + grid * width;               // behind the per-workgroup partials
  hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((unsigned)((width + 255) / 256), kRedGroups), dim3(256), 0, stream,
                     partial, grid, width, part2);
  hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, stream,
                     part2, width, (int64_t)m_out * k_in, grad_w, grad_b);
  SEGGER_LAUNCH_CHECK("wgrad_reduce_kernel");
  return SEGGER_OK;
}
}  // namespace
}  // namespace segger

extern "C" int segger_posmlp_wgrad(const void* dz1, int64_t ld_dz1, const float* pn, int64_t n_rows, float max_period,
                                   int32_t dtype, float* grad_w0, float* grad_b0, void* workspace, size_t workspace_bytes,
                                   segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  constexpr int M = 64, K = 256;
  SEGGER_REQUIRE(n_rows >= 0, "segger_posmlp_wgrad: negative size");
  SEGGER_REQUIRE(dtype == SEGGER_BF16 || dtype == SEGGER_F16, "segger_posmlp_wgrad: bf16 / f16 only");
  SEGGER_REQUIRE(grad_w0 != nullptr, "segger_posmlp_wgrad: grad_w0 is NULL");
  if (n_rows == 0) {
    SEGGER_HIP(hipMemsetAsync(grad_w0, 0, (size_t)M * K * sizeof(float), stream));
    if (grad_b0) SEGGER_HIP(hipMemsetAsync(grad_b0, 0, (size_t)M * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(dz1 && pn, "segger_posmlp_wgrad: NULL input");
  SEGGER_REQUIRE(aligned16(dz1) && aligned16(pn) && ld_dz1 >= M && (ld_dz1 * 2) % 16 == 0,
                 "segger_posmlp_wgrad: dz1 rows and pn must be 16-byte aligned");
  const size_t need = segger_linear_wgrad_workspace_bytes(n_rows, M, K);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_posmlp_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t grid = grid_for(n_rows, M, K);
  const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
  SEGGER_REQUIRE(((stages + grid - 1) / grid) * kStageRows * ld_dz1 * 2 < (int64_t)kOutOfRange,
                 "segger_posmlp_wgrad: a workgroup's row slab exceeds 1 GiB");
  WgradParams p{dz1, ld_dz1, nullptr, 0, n_rows, stages, (stages + grid - 1) / grid, static_cast<float*>(workspace), pn,
                logf(max_period)};
  if (dtype == SEGGER_BF16)
    hipLaunchKernelGGL((wgrad_kernel<bf16_t, M, K, 4, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((wgrad_kernel<f16_t, M, K, 4, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  SEGGER_LAUNCH_CHECK("wgrad_kernel (generated operand)");
  return reduce_partials(p.partial, grid, M, K, grad_w0, grad_b0, stream);

}

// grid of the fused positional backward: 2 workgroups per CU (VALU-bound on the regenerated features; every workgroup
// leaves an 82 KB partial), at least kMinStagesPerBlock stages each
static int64_t posmlp_bwd_grid(int64_t n_rows) {
  const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
  const int64_t want = (stages + kMinStagesPerBlock - 1) / kMinStagesPerBlock;
  const int64_t cap = 2 * kNumCu;
  return want < cap ? (want < 1 ? 1 : want) : cap;
}

extern "C" size_t segger_posmlp_bwd_pair_workspace_bytes(int64_t n_rows_a, int64_t n_rows_b) {
  const int64_t grid = (n_rows_a > 0 ? posmlp_bwd_grid(n_rows_a) : 0) + (n_rows_b > 0 ? posmlp_bwd_grid(n_rows_b) : 0);
  if (grid == 0) return 16;
  return (size_t)(grid + kRedGroups) * (size_t)(kPbWidth2 + kPbWidth0) * sizeof(float);
}
extern "C" size_t segger_posmlp_bwd_workspace_bytes(int64_t n_rows) { return segger_posmlp_bwd_pair_workspace_bytes(n_rows, 0); }

extern "C" int segger_posmlp_bwd_pair(const void* g_a, int64_t ld_ga, const void* z1_a, const float* pn_a, int64_t n_rows_a,
                                      const void* g_b, int64_t ld_gb, const void* z1_b, const float* pn_b, int64_t n_rows_b,
                                      const void* w2_t, float max_period, int32_t dtype, float* grad_w0, float* grad_b0,
                                      float* grad_w2, float* grad_b2, void* workspace, size_t workspace_bytes,
                                      segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_rows_a >= 0 && n_rows_b >= 0, "segger_posmlp_bwd: negative size");
  SEGGER_REQUIRE(dtype == SEGGER_BF16 || dtype == SEGGER_F16, "segger_posmlp_bwd: bf16 / f16 only");
  SEGGER_REQUIRE(grad_w0 && grad_b0 && grad_w2 && grad_b2, "segger_posmlp_bwd: NULL output");
  if (n_rows_a == 0 && n_rows_b == 0) {
    SEGGER_HIP(hipMemsetAsync(grad_w0, 0, (size_t)kPbD * kPbF * sizeof(float), stream));
    SEGGER_HIP(hipMemsetAsync(grad_b0, 0, (size_t)kPbD * sizeof(float), stream));
    SEGGER_HIP(hipMemsetAsync(grad_w2, 0, (size_t)kPbD * kPbD * sizeof(float), stream));
    SEGGER_HIP(hipMemsetAsync(grad_b2, 0, (size_t)kPbD * sizeof(float), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(w2_t && aligned16(w2_t), "segger_posmlp_bwd: NULL / misaligned W2^T");
  const size_t need = segger_posmlp_bwd_pair_workspace_bytes(n_rows_a, n_rows_b);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_posmlp_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const int64_t grid_a = n_rows_a > 0 ? posmlp_bwd_grid(n_rows_a) : 0, grid_b = n_rows_b > 0 ? posmlp_bwd_grid(n_rows_b) : 0;
  const int64_t grid = grid_a + grid_b;
  float* part2 = static_cast<float*>(workspace);
  float* part0 = part2 + (size_t)(grid + kRedGroups) * kPbWidth2;
  const float lmp = logf(max_period);
  auto side = [&](const void* g, int64_t ld_g, const void* z1, const float* pn, int64_t n_rows, int64_t grid_s, int64_t first,
                  PosBwdParams& out) -> int {
    if (n_rows == 0) { out = PosBwdParams{nullptr, kPbD, nullptr, nullptr, w2_t, 0, 0, 0, part2, part0, lmp}; return SEGGER_OK; }
    SEGGER_REQUIRE(g && z1 && pn, "segger_posmlp_bwd: NULL input");
    SEGGER_REQUIRE(aligned16(g) && aligned16(z1) && aligned16(pn) && ld_g >= kPbD && (ld_g * 2) % 16 == 0,
                   "segger_posmlp_bwd: rows must be 16-byte aligned");
    const int64_t stages = (n_rows + kStageRows - 1) / kStageRows;
    SEGGER_REQUIRE(((stages + grid_s - 1) / grid_s + kPbAhead) * kStageRows * ld_g * 2 < (int64_t)kOutOfRange,
                   "segger_posmlp_bwd: a workgroup's row slab exceeds 1 GiB");
    out = PosBwdParams{g, ld_g, z1, pn, w2_t, n_rows, stages, (stages + grid_s - 1) / grid_s, part2 + (size_t)first * kPbWidth2,
                       part0 + (size_t)first * kPbWidth0, lmp};
    return SEGGER_OK;
  };
  PosBwdParams pa, pb;
  int rc = side(g_a, ld_ga, z1_a, pn_a, n_rows_a, grid_a, 0, pa);
  if (rc != SEGGER_OK) return rc;
  rc = side(g_b, ld_gb, z1_b, pn_b, n_rows_b, grid_b, grid_a, pb);
  if (rc != SEGGER_OK) return rc;
  if (dtype == SEGGER_BF16)
    hipLaunchKernelGGL((posmlp_bwd_kernel<bf16_t>), dim3((unsigned)grid), dim3(512), 0, stream, pa, pb, (int)grid_a);
  else
    hipLaunchKernelGGL((posmlp_bwd_kernel<f16_t>), dim3((unsigned)grid), dim3(512), 0, stream, pa, pb, (int)grid_a);
  SEGGER_LAUNCH_CHECK("posmlp_bwd_kernel");
  rc = reduce_partials(part2, grid, kPbD, kPbD, grad_w2, grad_b2, stream);
  if (rc != SEGGER_OK) return rc;
  return reduce_partials(part0, grid, kPbD, kPbF, grad_w0, grad_b0, stream);
}

extern "C" int segger_posmlp_bwd(const void* g, int64_t ld_g, const void* z1, const float* pn, const void* w2_t,
                                 int64_t n_rows, float max_period, int32_t dtype, float* grad_w0, float* grad_b0,
                                 float* grad_w2, float* grad_b2, void* workspace, size_t workspace_bytes,
                                 segger_stream_t stream_) {
  return segger_posmlp_bwd_pair(g, ld_g, z1, pn, n_rows, nullptr, kPbD, nullptr, nullptr, 0, w2_t, max_period, dtype, grad_w0,
                                grad_b0, grad_w2, grad_b2, workspace, workspace_bytes, stream_);
}
