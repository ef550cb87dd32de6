// Tall-skinny projection GEMM on the matrix cores:  Y[n, M] = X[n, K] * W[M, K]^T (+ bias)
// for n ~ 10^6 node rows and small K, M (<= 384): the lin_l / lin_r / lin_last maps
// of the encoder and their data-gradients (dX = dY * W, called with W^T).
//
// The op is HBM-bound (2*n*(K+M) bytes against 2*n*K*M flops at K,M <= 384), so the
// design goal is "read X once, write Y once":
//   * one workgroup (4 waves) owns 128 consecutive rows and produces ALL M columns,
//     so X is never re-read per column tile;
//   * a wave keeps its 32 rows of X as MFMA B-operand fragments in registers for
//     the whole block (K/16 x 4 VGPRs), loaded straight from global memory;
//   * W is streamed through LDS in chunks of 64 output columns shared by the 4
//     waves (row stride K*2+16 B: conflict-free ds_read_b128 of A fragments), with
//     the next chunk prefetched into registers under the MFMAs;
//   * D = W_tile * X_tile^T  (v_mfma_f32_32x32x16): a lane owns one DATA ROW and
//     4-column groups, so the epilogue packs 4 bf16 -> ds_write_b64 into a wave-private
//     LDS tile and streams it out as full 128-byte row segments (16 B per lane).
#include "common.h"

namespace segger {
namespace {

typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRowsPerBlock = 128;
constexpr int kChunk = 64;             // output columns per W chunk

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

struct LinearParams {
  const void* x; int64_t ldx;
  const void* w;            // [m_out, K] row-major, same dtype as x
  const float* bias;        // [m_out] or NULL
  void* y; int64_t ldy;
  int64_t n_rows;
  int m_out;
  // optional per-row additive term: y[row, :] += rowbias[rowidx[row], :] (fp32 table, row stride ld_rb elements) -- the
  // part of a projection that depends on a row only through a small categorical id (ist_encoder's gene embedding)
  const void* rowbias; const int32_t* rowidx; int64_t ld_rb;      // (table in the activation dtype)
  // optional epilogue factor: y[row, c] *= silu'(gate[row, c]) (gate in the activation dtype, row stride ld_gate) -- the
  // backward of Linear -> SiLU without a separate elementwise pass over dX
  const void* gate; int64_t ld_gate;
};

template <int K> constexpr int linear_lds_bytes() { return kChunk * (K * 2 + 16) + 4 * 32 * (kChunk * 2 + 16); }

// `lds`: the workgroup's shared array of linear_lds_bytes<K>() bytes (declared by the kernel: two bodies of a pair launch
// share one array instead of adding theirs up)
template <typename T, int K, bool RB = false, bool SG = false>
__device__ __forceinline__ void linear_fwd_body(const LinearParams p, const int64_t bid, unsigned char* lds) {
  constexpr int NK = K / 16;                       // k-steps
  constexpr int WSTRIDE = K * 2 + 16;              // bytes per LDS row of W
  constexpr int ESTRIDE = kChunk * 2 + 16;         // bytes per LDS row of the epilogue tile
  constexpr int PIECES = kChunk * K * 2 / 16;      // 16-byte pieces per W chunk
  constexpr int PPT = PIECES / 256;                // pieces per thread
  static_assert(PIECES % 256 == 0, "chunk must split evenly over the block");
  unsigned char* lds_w = lds;
  unsigned char* lds_e = lds + kChunk * WSTRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = bid * kRowsPerBlock + wave * 32;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const T* __restrict__ w = static_cast<const T*>(p.w);
  T* __restrict__ y = static_cast<T*>(p.y);

  // ---- X fragments: lane (r,h) holds X[row0+r][16s+8h .. +8) for every k-step s -----------------
  u32x4 xb[NK];
  {
    int64_t row = row0 + r;
    if (row >= p.n_rows) row = p.n_rows - 1;       // clamp: loaded, never stored
    const T* xr = x + row * p.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < NK; ++s) xb[s] = *reinterpret_cast<const u32x4*>(xr + 16 * s);
  }

  // ---- W chunk staging ------------------------------------------------------------------------------
  u32x4 wreg[PPT];
  auto w_fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int piece = tid + 256 * i;
      const int wrow = piece / (K / 8), wcol = piece % (K / 8);
      wreg[i] = *reinterpret_cast<const u32x4*>(w + (int64_t)(c0 + wrow) * K + wcol * 8);
    }
  };
  auto w_commit = [&]() {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int piece = tid + 256 * i;
      const int wrow = piece / (K / 8), wcol = piece % (K / 8);
      *reinterpret_cast<u32x4*>(lds_w + wrow * WSTRIDE + wcol * 16) = wreg[i];
    }
  };

  const int n_chunks = p.m_out / kChunk;
  // per-row additive table row of this lane's data row (its columns are fetched per chunk, ahead of the MFMAs)
  const T* gate_row = nullptr;
  // RB: the table rows of this wave's 32 data rows are fetched as whole 128-byte chunk segments -- lane (8 i + lane / 8,
  // piece lane % 8) loads 16 bytes, 8 rows x 128 B per instruction -- and handed to the lanes that own the output
  // columns through the wave's epilogue tile.  (Round 2 let every lane fetch its own 8-byte column groups: 32 table
  // rows x 16 B per instruction, twice the instructions and four times the cache lines per instruction: 0.31 ms for
  // the first layer against 0.22 ms for the plain projection.)
  const T* rb_rows[4] = {nullptr, nullptr, nullptr, nullptr};
  if (RB) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row = row0 + 8 * i + (lane >> 3);
      if (row >= p.n_rows) row = p.n_rows - 1;
      rb_rows[i] = static_cast<const T*>(p.rowbias) + (int64_t)p.rowidx[row] * p.ld_rb + 8 * (lane & 7);
    }
  }
  if (SG) {
    int64_t row = row0 + r;
    if (row >= p.n_rows) row = p.n_rows - 1;
    gate_row = static_cast<const T*>(p.gate) + row * p.ld_gate + 4 * h;
  }
  w_fetch(0);
  for (int c = 0; c < n_chunks; ++c) {
    const int c0 = c * kChunk;
    w_commit();                                    // all waves left the previous chunk's MFMA loop (epilogue barrier)
    __syncthreads();
    if (c + 1 < n_chunks) w_fetch(c0 + kChunk);    // prefetch under the MFMAs
    // this lane's 4-column groups of the table row / the gate row, requested ahead of the MFMAs
    uint2 rbv[SG ? 2 : 1][SG ? 4 : 1];
    u32x4 rbq[RB ? 4 : 1];
    if (SG) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) rbv[ct][g] = *reinterpret_cast<const uint2*>(gate_row + c0 + ct * 32 + 8 * g);
    }
    if (RB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rbq[i] = *reinterpret_cast<const u32x4*>(rb_rows[i] + c0);
    }

    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < NK; ++s) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(lds_w + (ct * 32 + r) * WSTRIDE + (16 * s + 8 * h) * 2);
        acc[ct] = Mfma<T>::run(a, xb[s], acc[ct]);
      }
    }

    // ---- epilogue: lane owns data row r and output columns ct*32 + 8g + 4h + {0..3} -------------
    unsigned char* et = lds_e + wave * 32 * ESTRIDE;
    if (RB) {
      // table segments -> the wave's tile (same layout as the output tile); every lane then reads its own column
      // groups back and overwrites exactly those bytes with the result: LDS operations of one wave execute in order
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<u32x4*>(et + (8 * i + (lane >> 3)) * ESTRIDE + (lane & 7) * 16) = rbq[i];
    }

#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = ct * 32 + 8 * g + 4 * h;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ct][4 * g + j] + (p.bias ? p.bias[c0 + col + j] : 0.f);
        if (RB) {
          const uint2 tt = *reinterpret_cast<const uint2*>(et + r * ESTRIDE + col * 2);
          float t4[4];
          Vec8<T>::unpack2(tt.x, t4[0], t4[1]);
          Vec8<T>::unpack2(tt.y, t4[2], t4[3]);
          v[0] += t4[0]; v[1] += t4[1]; v[2] += t4[2]; v[3] += t4[3];
        }
        if (SG) {
          const uint2 zz = rbv[SG ? ct : 0][SG ? g : 0];
          float z[4];
          Vec8<T>::unpack2(zz.x, z[0], z[1]);
          Vec8<T>::unpack2(zz.y, z[2], z[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z[j]));
            v[j] *= sg * (1.0f + z[j] * (1.0f - sg));         // d/dz [z sigmoid(z)]
          }
        }
        uint2 pk;
        pk.x = Vec8<T>::pack(v[0], v[1]);
        pk.y = Vec8<T>::pack(v[2], v[3]);
        *reinterpret_cast<uint2*>(et + r * ESTRIDE + col * 2) = pk;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int er = 8 * i + (lane >> 3), piece = lane & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(et + er * ESTRIDE + piece * 16);
      const int64_t row = row0 + er;
      if (row < p.n_rows) *reinterpret_cast<u32x4*>(y + row * p.ldy + c0 + piece * 8) = v;
    }
  }
}

template <typename T, int K, bool RB = false, bool SG = false>
__global__ __launch_bounds__(256, 2) void linear_fwd_kernel(LinearParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[linear_lds_bytes<K>()];
  linear_fwd_body<T, K, RB, SG>(p, blockIdx.x, lds);
}

// Two projections with the same K in ONE launch: the blocks of `b` (a few: the boundary side of a hetero layer, ~500 rows)
// are dispatched first and run beside the blocks of `a` instead of as a 7 us launch of their own (a captured 1M-edge
// training step has six such pairs; same idea as gatv2_fwd_pair_kernel).
template <typename T, int KA, int KB = KA>
__global__ __launch_bounds__(256, 2) void linear_fwd_pair_kernel(LinearParams a, LinearParams b, int nb_b) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[linear_lds_bytes<(KA > KB ? KA : KB)>()];
  if ((int)blockIdx.x < nb_b) linear_fwd_body<T, KB>(b, blockIdx.x, lds);
  else linear_fwd_body<T, KA>(a, (int64_t)blockIdx.x - nb_b, lds);
}

// ---- resident-W form for large row counts ----------------------------------------------------------------------
// The chunked kernel above keeps four 128-row blocks per CU, each of which lives ~25 us for ~3 us of matrix work: two
// barriers per 64-column chunk, W re-committed to LDS per chunk and block, the X fragments loaded with nothing to
// overlap.  Bytes in flight per CU / block lifetime is what bounds it (4.3 TB/s of stores with the loads taken out,
// 1.97 TB/s of loads with the stores taken out: `EXP_NOLOAD` / `EXP_NOSTORE` builds, round 3).  For n >= ~10^5 rows
// and K * M small enough, ONE persistent workgroup per CU holds the whole W in LDS (384 x 128 bf16 = 102 KB with the
// conflict-free row stride) and its 12 waves walk 32-row tiles on their own: no barrier after the W load, the next
// tile's X fragments requested before the current tile's MFMAs, the epilogue tile wave-private.  NCH (chunks of 64
// output columns) is a template parameter so that the chunk loop is straight-line code: the wait for the prefetched
// X fragments is then vmcnt(4 * NCH) -- the tile's output stores stay in flight (gfx9 counts stores in vmcnt).
constexpr int kResWaves = 12;
constexpr int kResMaxOut = 384;

template <typename T, int K, int NCH>
__global__ __launch_bounds__(kResWaves * 64, 1) void linear_res_kernel(LinearParams p, int64_t n_tiles) {
  constexpr int NK = K / 16;
  constexpr int WSTRIDE = K * 2 + 16;
  constexpr int ESTRIDE = kChunk * 2 + 16;
  constexpr int M = NCH * kChunk;
  __shared__ __attribute__((aligned(16))) unsigned char lds[M * WSTRIDE + kResWaves * 32 * ESTRIDE];
  __shared__ __attribute__((aligned(16))) float lds_bias[M];
  unsigned char* lds_w = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* et = lds + M * WSTRIDE + wave * 32 * ESTRIDE;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const T* __restrict__ w = static_cast<const T*>(p.w);
  T* __restrict__ y = static_cast<T*>(p.y);

  for (int piece = tid; piece < M * K / 8; piece += kResWaves * 64) {
    const int wrow = piece / (K / 8), wcol = piece % (K / 8);
    *reinterpret_cast<u32x4*>(lds_w + wrow * WSTRIDE + wcol * 16) = *reinterpret_cast<const u32x4*>(w + (int64_t)wrow * K + wcol * 8);
  }
  for (int i = tid; i < M; i += kResWaves * 64) lds_bias[i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();

  const int64_t stride = (int64_t)gridDim.x * kResWaves;
  int64_t tile = (int64_t)blockIdx.x * kResWaves + wave;          // at any time the grid covers one contiguous run of tiles
  const int64_t n_full = p.n_rows / 32;                           // tiles without a missing row

  auto x_load = [&](u32x4 (&dst)[NK], int64_t t) {
    int64_t row = t * 32 + r;
    if (row >= p.n_rows) row = p.n_rows - 1;                      // clamp: loaded, never stored
    const T* xr = x + row * p.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < NK; ++s) dst[s] = *reinterpret_cast<const u32x4*>(xr + 16 * s);
  };

  u32x4 xb[NK], xn[NK];
  if (tile < n_tiles) x_load(xn, tile);

  auto do_tile = [&](auto full_c, const int64_t t) {
    constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
    for (int s = 0; s < NK; ++s) xb[s] = xn[s];
    if (t + stride < n_tiles) x_load(xn, t + stride);             // under this tile's MFMAs
    const int64_t row0 = t * 32;
    // NCH == 2 (M = 128): fully unrolled the two chunks' MFMA chains and epilogues are interleaved by the scheduler and the
    // tile loop spills 136 bytes per lane (7 reloads inside it); kept as a loop the kernel needs no scratch
#pragma unroll NCH == 2 ? 1 : NCH
    for (int c = 0; c < NCH; ++c) {
      const int c0 = c * kChunk;
      f32x16 acc[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < NK; ++s) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(lds_w + (c0 + ct * 32 + r) * WSTRIDE + (16 * s + 8 * h) * 2);
          acc[ct] = Mfma<T>::run(a, xb[s], acc[ct]);
        }
      }
      // epilogue through the wave's own LDS tile (LDS operations of one wave execute in order: no barrier)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = ct * 32 + 8 * g + 4 * h;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_bias + c0 + col);
          uint2 pk;
          pk.x = Vec8<T>::pack(acc[ct][4 * g + 0] + bv.x, acc[ct][4 * g + 1] + bv.y);
          pk.y = Vec8<T>::pack(acc[ct][4 * g + 2] + bv.z, acc[ct][4 * g + 3] + bv.w);
          *reinterpret_cast<uint2*>(et + r * ESTRIDE + col * 2) = pk;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int er = 8 * i + (lane >> 3), piece = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(et + er * ESTRIDE + piece * 16);
        const int64_t row = row0 + er;
        if (FULL || row < p.n_rows) *reinterpret_cast<u32x4*>(y + row * p.ldy + c0 + piece * 8) = v;
      }
    }
  };
  // full tiles: unconditional stores (events the wait-count analysis can count on); the one partial tile afterwards
#pragma unroll 1
  for (; tile < n_full; tile += stride) do_tile(std::true_type{}, tile);
  if (tile < n_tiles) do_tile(std::false_type{}, tile);
}

template <typename T>
int launch_linear_res(const LinearParams& p, int k_in, hipStream_t stream, bool* done) {
  *done = false;
  const int n_cu = device_cu_count();           // per device (common.h)
  const int64_t n_tiles = (p.n_rows + 31) / 32;
  // worth it only when every wave of the persistent grid gets a few tiles
  if (n_cu <= 0 || p.rowbias || p.gate || n_tiles < (int64_t)n_cu * kResWaves * 2) return SEGGER_OK;
  dim3 grid((unsigned)n_cu), block(kResWaves * 64);
#define RES_CASE(KK, NN) \
  if (k_in == KK && p.m_out == NN * kChunk) { \
    hipLaunchKernelGGL((linear_res_kernel<T, KK, NN>), grid, block, 0, stream, p, n_tiles); \
    SEGGER_LAUNCH_CHECK("linear_res_kernel"); *done = true; return SEGGER_OK; }
  RES_CASE(128, 6) RES_CASE(128, 2) RES_CASE(128, 1) RES_CASE(64, 6) RES_CASE(64, 2) RES_CASE(64, 1)
#undef RES_CASE
  return SEGGER_OK;
}

template <typename T>
int launch_linear(const LinearParams& p, int k_in, hipStream_t stream) {
  bool done = false;
  const int rc = launch_linear_res<T>(p, k_in, stream, &done);
  if (rc != SEGGER_OK || done) return rc;
  const int64_t nb = (p.n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
  if (nb > 0x7fffffffLL) { set_error("segger_linear_fwd: too many rows"); return SEGGER_EUNSUPPORTED; }
  dim3 grid((unsigned)nb), block(256);
  if (p.gate) {
    if (k_in != 64) {
      set_error("segger_linear_fwd_silu_grad: k_in=%d not supported (64)", k_in);
      return SEGGER_EUNSUPPORTED;
    }
    hipLaunchKernelGGL((linear_fwd_kernel<T, 64, false, true>), grid, block, 0, stream, p);
    SEGGER_LAUNCH_CHECK("linear_fwd_kernel (silu')");
    return SEGGER_OK;
  }
  if (p.rowbias) {
    switch (k_in) {
      case 64:  hipLaunchKernelGGL((linear_fwd_kernel<T, 64, true>), grid, block, 0, stream, p); break;
      case 128: hipLaunchKernelGGL((linear_fwd_kernel<T, 128, true>), grid, block, 0, stream, p); break;
      default:
        set_error("segger_linear_fwd_rowbias: k_in=%d not supported (64, 128)", k_in);
        return SEGGER_EUNSUPPORTED;
    }
    SEGGER_LAUNCH_CHECK("linear_fwd_kernel (row bias)");
    return SEGGER_OK;
  }
  switch (k_in) {
    case 64:  hipLaunchKernelGGL((linear_fwd_kernel<T, 64>), grid, block, 0, stream, p); break;
    case 128: hipLaunchKernelGGL((linear_fwd_kernel<T, 128>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((linear_fwd_kernel<T, 256>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((linear_fwd_kernel<T, 384>), grid, block, 0, stream, p); break;
    default:
      set_error("segger_linear_fwd: k_in=%d not supported (64, 128, 256, 384)", k_in);
      return SEGGER_EUNSUPPORTED;
  }
  SEGGER_LAUNCH_CHECK("linear_fwd_kernel");
  return SEGGER_OK;
}

template <typename T>
int launch_linear_pair(const LinearParams& a, const LinearParams& b, int k_in, hipStream_t stream, int k_b = 0) {
  if (k_b == 0) k_b = k_in;
  bool done = false;
  int rc = launch_linear_res<T>(a, k_in, stream, &done);       // a large `a` goes its own (persistent) way
  if (rc != SEGGER_OK) return rc;
  if (done) return launch_linear<T>(b, k_b, stream);
  if (k_b != k_in) {
    // different input widths (a first layer's two data gradients: 384 and 128 columns of dY): the pairs instantiated
    const int64_t na = (a.n_rows + kRowsPerBlock - 1) / kRowsPerBlock, nb = (b.n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    if (k_in == 384 && k_b == 128 && na > 0 && nb > 0 && na + nb <= 0x7fffffffLL) {
      hipLaunchKernelGGL((linear_fwd_pair_kernel<T, 384, 128>), dim3((unsigned)(na + nb)), dim3(256), 0, stream, a, b, (int)nb);
      SEGGER_LAUNCH_CHECK("linear_fwd_pair_kernel");
      return SEGGER_OK;
    }
    rc = launch_linear<T>(a, k_in, stream);
    return rc != SEGGER_OK ? rc : launch_linear<T>(b, k_b, stream);
  }
  const int64_t nb_a = (a.n_rows + kRowsPerBlock - 1) / kRowsPerBlock, nb_b = (b.n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
  if (nb_a + nb_b > 0x7fffffffLL) { set_error("segger_linear_fwd_pair: too many rows"); return SEGGER_EUNSUPPORTED; }
  if (nb_a == 0) return launch_linear<T>(b, k_in, stream);
  if (nb_b == 0) return launch_linear<T>(a, k_in, stream);
  dim3 grid((unsigned)(nb_a + nb_b)), block(256);
  switch (k_in) {
    case 64:  hipLaunchKernelGGL((linear_fwd_pair_kernel<T, 64>), grid, block, 0, stream, a, b, (int)nb_b); break;
    case 128: hipLaunchKernelGGL((linear_fwd_pair_kernel<T, 128>), grid, block, 0, stream, a, b, (int)nb_b); break;
    case 256: hipLaunchKernelGGL((linear_fwd_pair_kernel<T, 256>), grid, block, 0, stream, a, b, (int)nb_b); break;
    case 384: hipLaunchKernelGGL((linear_fwd_pair_kernel<T, 384>), grid, block, 0, stream, a, b, (int)nb_b); break;
    default:
      set_error("segger_linear_fwd_pair: k_in=%d not supported (64, 128, 256, 384)", k_in);
      return SEGGER_EUNSUPPORTED;
  }
  SEGGER_LAUNCH_CHECK("linear_fwd_pair_kernel");
  return SEGGER_OK;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" int segger_linear_fwd_pair(const segger_linear_args* a, const segger_linear_args* b, int32_t k_in, int32_t dtype,
                                      segger_stream_t stream) {
  return segger_linear_fwd_pair_k(a, k_in, b, k_in, dtype, stream);
}

extern "C" int segger_linear_fwd_pair_k(const segger_linear_args* a, int32_t k_a, const segger_linear_args* b, int32_t k_b,
                                        int32_t dtype, segger_stream_t stream) {
  SEGGER_REQUIRE(a && b, "segger_linear_fwd_pair: NULL args");
  const segger_linear_args* both[2] = {a, b};
  const int32_t ks[2] = {k_a, k_b};
  if (dtype == SEGGER_F32) {      // fp32 storage: two launches of the exact-fp32 kernel
    for (int i = 0; i < 2; ++i) {
      const segger_linear_args* q = both[i];
      const int rc = segger_linear_fwd(q->x, q->ldx, q->w, q->bias, q->y, q->ldy, q->n_rows, ks[i], q->m_out, dtype, stream);
      if (rc != SEGGER_OK) return rc;
    }
    return SEGGER_OK;
  }
  LinearParams p[2];
  for (int i = 0; i < 2; ++i) {
    const segger_linear_args* q = both[i];
    const int32_t k_in = ks[i];
    SEGGER_REQUIRE(q->n_rows >= 0 && q->m_out > 0, "segger_linear_fwd_pair: bad sizes");
    if (!segger_linear_supported(k_in, q->m_out, dtype)) {
      set_error("segger_linear_fwd_pair: k_in=%d m_out=%d dtype=%d not supported", k_in, q->m_out, dtype);
      return SEGGER_EUNSUPPORTED;
    }
    SEGGER_REQUIRE(q->n_rows == 0 || (q->x && q->w && q->y), "segger_linear_fwd_pair: NULL pointer");
    SEGGER_REQUIRE(aligned16(q->x) && aligned16(q->w) && aligned16(q->y), "segger_linear_fwd_pair: pointers must be 16-byte aligned");
    SEGGER_REQUIRE(q->ldx >= k_in && q->ldy >= q->m_out && (q->ldx * 2) % 16 == 0 && (q->ldy * 2) % 16 == 0,
                   "segger_linear_fwd_pair: bad leading dimension");
    p[i] = LinearParams{q->x, q->ldx, q->w, q->bias, q->y, q->ldy, q->n_rows, q->m_out, nullptr, nullptr, 0, nullptr, 0};
  }
  return dtype == SEGGER_BF16 ? launch_linear_pair<bf16_t>(p[0], p[1], k_a, (hipStream_t)stream, k_b)
                              : launch_linear_pair<f16_t>(p[0], p[1], k_a, (hipStream_t)stream, k_b);
}

extern "C" int segger_linear_supported(int32_t k_in, int32_t m_out, int32_t dtype) {
  const bool k_ok = k_in == 64 || k_in == 128 || k_in == 256 || k_in == 384;
  return k_ok && m_out > 0 && m_out % kChunk == 0 && (dtype == SEGGER_BF16 || dtype == SEGGER_F16 || dtype == SEGGER_F32);
}

extern "C" int segger_linear_fwd(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy,
                                 int64_t n_rows, int32_t k_in, int32_t m_out, int32_t dtype, segger_stream_t stream) {
  return segger_linear_fwd_rowbias(x, ldx, w, bias, nullptr, 0, nullptr, y, ldy, n_rows, k_in, m_out, dtype, stream);
}

extern "C" int segger_linear_fwd_silu_grad(const void* x, int64_t ldx, const void* w, const void* gate, int64_t ld_gate,
                                           void* y, int64_t ldy, int64_t n_rows, int32_t k_in, int32_t m_out,
                                           int32_t dtype, segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows >= 0 && k_in > 0 && m_out > 0, "segger_linear_fwd_silu_grad: bad sizes");
  if (n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(segger_linear_supported(k_in, m_out, dtype) && k_in == 64 && dtype != SEGGER_F32,
                 "segger_linear_fwd_silu_grad: k_in 64, m_out %% 64 == 0, bf16 / f16");
  SEGGER_REQUIRE(x && w && y && gate, "segger_linear_fwd_silu_grad: NULL pointer");
  SEGGER_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(gate),
                 "segger_linear_fwd_silu_grad: pointers must be 16-byte aligned");
  SEGGER_REQUIRE(ldx >= k_in && ldy >= m_out && ld_gate >= m_out && (ldx * 2) % 16 == 0 && (ldy * 2) % 16 == 0 &&
                     (ld_gate * 2) % 8 == 0, "segger_linear_fwd_silu_grad: bad leading dimension");
  LinearParams p{x, ldx, w, nullptr, y, ldy, n_rows, m_out, nullptr, nullptr, 0, gate, ld_gate};
  return dtype == SEGGER_BF16 ? launch_linear<bf16_t>(p, k_in, (hipStream_t)stream)
                              : launch_linear<f16_t>(p, k_in, (hipStream_t)stream);
}

extern "C" int segger_linear_fwd_rowbias(const void* x, int64_t ldx, const void* w, const float* bias,
                                         const void* rowbias, int64_t ld_rb, const int32_t* rowidx, void* y, int64_t ldy,
                                         int64_t n_rows, int32_t k_in, int32_t m_out, int32_t dtype,
                                         segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows >= 0 && k_in > 0 && m_out > 0, "segger_linear_fwd: bad sizes");
  if (n_rows == 0) return SEGGER_OK;
  if (!segger_linear_supported(k_in, m_out, dtype)) {
    set_error("segger_linear_fwd: k_in=%d m_out=%d dtype=%d not supported (k_in in {64,128,256,384}, "
              "m_out %% 64 == 0, bf16/f16)", k_in, m_out, dtype);
    return SEGGER_EUNSUPPORTED;
  }
  SEGGER_REQUIRE(x && w && y, "segger_linear_fwd: NULL pointer");
  SEGGER_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), "segger_linear_fwd: pointers must be 16-byte aligned");
  if (dtype == SEGGER_F32) {       // fp32 storage: exact-fp32 MFMA kernel (csrc/linear_f32.hip); plain projection only
    SEGGER_REQUIRE(!rowbias == !rowidx, "segger_linear_fwd_rowbias: rowbias and rowidx go together");
    SEGGER_REQUIRE(!rowbias || (aligned16(rowbias) && ld_rb >= m_out && ld_rb % 4 == 0),
                   "segger_linear_fwd_rowbias: the fp32 table needs 16-byte aligned rows of at least m_out elements");
    SEGGER_REQUIRE(ldx >= k_in && ldy >= m_out && ldx % 4 == 0 && ldy % 4 == 0 && (!bias || aligned16(bias)),
                   "segger_linear_fwd: fp32 rows (and the bias) must be 16-byte aligned");
    return linear_f32_launch(x, ldx, w, bias, y, ldy, n_rows, k_in, m_out, (hipStream_t)stream,
                             static_cast<const float*>(rowbias), rowidx, ld_rb);
  }
  SEGGER_REQUIRE(ldx >= k_in && ldy >= m_out && (ldx * 2) % 16 == 0 && (ldy * 2) % 16 == 0,
                 "segger_linear_fwd: bad leading dimension");
  SEGGER_REQUIRE(!rowbias == !rowidx, "segger_linear_fwd_rowbias: rowbias and rowidx go together");
  SEGGER_REQUIRE(!rowbias || (aligned16(rowbias) && ld_rb >= m_out && ld_rb % 8 == 0),
                 "segger_linear_fwd_rowbias: the table needs 16-byte aligned rows of at least m_out elements");
  LinearParams p{x, ldx, w, bias, y, ldy, n_rows, m_out, rowbias, rowidx, ld_rb, nullptr, 0};
  return dtype == SEGGER_BF16 ? launch_linear<bf16_t>(p, k_in, (hipStream_t)stream)
                              : launch_linear<f16_t>(p, k_in, (hipStream_t)stream);
}
