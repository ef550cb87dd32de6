// The projections of the encoder at the REFERENCE'S arithmetic width: fp32 storage, fp32 products, fp32 accumulation
// (segger trains with a default Trainer(), i.e. fp32: src/segger/cli/segment.py:400-405), on the matrix cores'
// f32-input form v_mfma_f32_32x32x2_f32 -- exact fp32 (the same result as a chain of fmaf), 64 FLOP/clk/SIMD, i.e. 157
// TFLOP/s on the chip.  At n ~ 10^6 rows and K, M <= 384 these GEMMs are MFMA-bound (0.1 TFLOP against 2 GB: 0.63 ms of
// matrix time, 0.4 ms of traffic), unlike their bf16 forms, so the kernels below are built to keep the MFMA pipe fed and
// to touch every matrix once; the vendor GEMM they replace re-reads the tall operand per column macro-tile.
//
//   segger_linear_fwd (dtype f32):   Y[n, M] = X[n, K] W[M, K]^T + b            (also dX = dY W, called with W^T)
//   segger_linear_wgrad (dtype f32): dW[M, K] = dY[n, M]^T X[n, K],  db = sum_n dY
//
// Operand layout of v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31],
// ONE float each.  The k order of a product is free as long as A and B agree, so a lane fetches FOUR consecutive k of its
// row with one 16-byte load (lane half h covers k = 8 t + 4 h + u, u = 0..3) and feeds them to four MFMAs.
#include "common.h"

namespace segger {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ forward / dX
struct LinF32Params {
  const float* x; int64_t ldx;
  const float* w;            // [m_out, K] row-major
  const float* bias;         // [m_out] or NULL
  float* y; int64_t ldy;
  int64_t n_rows;
  int m_out;
  // optional per-row additive term: y[row, :] += rowbias[rowidx[row], :] (fp32 table, row stride ld_rb): the per-gene
  // table form of the first layer (ops.embed_linear), as in linear.hip
  const float* rowbias; const int32_t* rowidx; int64_t ld_rb;
  // optional: y[row, c] *= act'(gate[row, c]) (gate_kind 1 = GELU, 2 = SiLU): the data gradient through an activation,
  // torch's separate gelu_backward / silu_backward pass folded into the epilogue (as segger_linear_fwd_silu_grad at 16 bit)
  const float* gate; int64_t ld_gate; int gate_kind;
  // optional second output: y_act[row, c] = act(y[row, c]) (act_kind 1 = GELU, 2 = SiLU) from the same epilogue -- the
  // pre-activation (kept for the backward) and the activation (the next layer's input) from one kernel
  float* y_act; int64_t ld_yact; int act_kind;
};

// A workgroup (4 waves) owns 128 rows and produces all M columns: X is read once.  A wave keeps its 32 rows as B-operand
// fragments in registers (K / 2 VGPRs); W streams through LDS in chunks of CH output columns (row stride K + 4 floats:
// the ds_read_b128 of 16 different rows hit 16 disjoint 4-bank groups), the next chunk prefetched under the MFMAs.
// D = W_chunk X_tile^T: a lane owns one DATA row and 4-column groups of the output -> 16-byte stores, 32 bytes per row
// and instruction, the four groups of a tile completing 128-byte lines.
template <int K>
__global__ __launch_bounds__(256, K <= 256 ? 2 : 1) void linear_f32_kernel(LinF32Params p) {
  constexpr int CH = K <= 128 ? 64 : 32;                 // output columns per W chunk
  constexpr int CT = CH / 32;                            // 32-column tiles per chunk
  constexpr int NT = K / 8;                              // 16-byte k groups
  constexpr int WS = K + 4;                              // LDS row stride (floats)
  constexpr int PPT = CH * K / 4 / 256;                  // float4 pieces of a chunk per thread
  static_assert((CH * K / 4) % 256 == 0, "chunk must split evenly over the block");
  __shared__ __attribute__((aligned(16))) float lds_w[CH * WS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
  int64_t row = row0 + r;
  const bool row_ok = row < p.n_rows;
  if (!row_ok) row = p.n_rows - 1;                       // clamp: loaded, never stored

  f32x4 xq[NT];
  {
    const float* xr = p.x + row * p.ldx + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t) xq[t] = *reinterpret_cast<const f32x4*>(xr + 8 * t);
  }
  f32x4 wreg[PPT];
  auto w_fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int piece = tid + 256 * i, wrow = piece / (K / 4), wcol = piece % (K / 4);
      wreg[i] = *reinterpret_cast<const f32x4*>(p.w + (int64_t)(c0 + wrow) * K + wcol * 4);
    }
  };
  auto w_commit = [&]() {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int piece = tid + 256 * i, wrow = piece / (K / 4), wcol = piece % (K / 4);
      *reinterpret_cast<f32x4*>(lds_w + wrow * WS + wcol * 4) = wreg[i];
    }
  };

  const int n_chunks = p.m_out / CH;
  w_fetch(0);
  for (int c = 0; c < n_chunks; ++c) {
    const int c0 = c * CH;
    __syncthreads();                                     // every wave has left the previous chunk's reads
    w_commit();
    __syncthreads();
    if (c + 1 < n_chunks) w_fetch(c0 + CH);
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const f32x4 wq = *reinterpret_cast<const f32x4*>(lds_w + (ct * 32 + r) * WS + 8 * t + 4 * h);
        acc[ct] = mfma_f32(wq.x, xq[t].x, acc[ct]);
        acc[ct] = mfma_f32(wq.y, xq[t].y, acc[ct]);
        acc[ct] = mfma_f32(wq.z, xq[t].z, acc[ct]);
        acc[ct] = mfma_f32(wq.w, xq[t].w, acc[ct]);
      }
    }
    if (row_ok) {
      float* yr = p.y + row * p.ldy + c0;
      const float* tr = p.rowbias ? p.rowbias + (int64_t)p.rowidx[row] * p.ld_rb + c0 : nullptr;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = ct * 32 + 8 * g + 4 * h;
          f32x4 v = f32x4{acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
          if (p.bias) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + c0 + col);
            v = v + b;
          }
          if (tr) v = v + *reinterpret_cast<const f32x4*>(tr + col);
          if (p.gate) {
            const f32x4 gq = *reinterpret_cast<const f32x4*>(p.gate + row * p.ld_gate + c0 + col);
            v = f32x4{v.x * gate_grad(gq.x, p.gate_kind), v.y * gate_grad(gq.y, p.gate_kind),
                      v.z * gate_grad(gq.z, p.gate_kind), v.w * gate_grad(gq.w, p.gate_kind)};
          }
          *reinterpret_cast<f32x4*>(yr + col) = v;
          if (p.y_act)
            *reinterpret_cast<f32x4*>(p.y_act + row * p.ld_yact + c0 + col) =
                f32x4{act_apply(v.x, p.act_kind), act_apply(v.y, p.act_kind), act_apply(v.z, p.act_kind), act_apply(v.w, p.act_kind)};
        }
      }
    }
  }
}

// The data gradients dX[n, M] = dY[n, K = 3 HC] W^T have a LONG reduction and few output columns: holding all K of a row
// in registers (192 VGPRs) leaves one wave per SIMD.  This form walks K in slices of KS instead and keeps the whole
// [32 rows, M] output of a wave in accumulators (M / 32 tiles): per slice a wave loads KS / 2 registers of its rows and
// the workgroup stages W[:, slice] ([M, KS], row stride KS + 4) in LDS; the output is stored once at the end.
template <int K, int M>
__global__ __launch_bounds__(256, 2) void linear_f32_kslice_kernel(LinF32Params p) {
  constexpr int KS = M <= 128 ? 128 : 32;
  constexpr int NS = K / KS, NT = KS / 8, MT = M / 32, WS = KS + 4;
  constexpr int PPT = M * KS / 4 / 256;
  static_assert(K % KS == 0 && (M * KS / 4) % 256 == 0, "slices must tile the operands");
  __shared__ __attribute__((aligned(16))) float lds_w[M * WS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + r;
  const bool row_ok = row < p.n_rows;
  if (!row_ok) row = p.n_rows - 1;
  const float* xr = p.x + row * p.ldx + 4 * h;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
  for (int sl = 0; sl < NS; ++sl) {
    f32x4 xq[NT], wreg[PPT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xq[t] = *reinterpret_cast<const f32x4*>(xr + sl * KS + 8 * t);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int piece = tid + 256 * i, wrow = piece / (KS / 4), wcol = piece % (KS / 4);
      wreg[i] = *reinterpret_cast<const f32x4*>(p.w + (int64_t)wrow * K + sl * KS + wcol * 4);
    }
    __syncthreads();                                     // the previous slice's reads are done
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int piece = tid + 256 * i, wrow = piece / (KS / 4), wcol = piece % (KS / 4);
      *reinterpret_cast<f32x4*>(lds_w + wrow * WS + wcol * 4) = wreg[i];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 wq = *reinterpret_cast<const f32x4*>(lds_w + (mt * 32 + r) * WS + 8 * t + 4 * h);
        acc[mt] = mfma_f32(wq.x, xq[t].x, acc[mt]);
        acc[mt] = mfma_f32(wq.y, xq[t].y, acc[mt]);
        acc[mt] = mfma_f32(wq.z, xq[t].z, acc[mt]);
        acc[mt] = mfma_f32(wq.w, xq[t].w, acc[mt]);
      }
    }
  }
  if (row_ok) {
    float* yr = p.y + row * p.ldy;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = mt * 32 + 8 * g + 4 * h;
        f32x4 v = f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
        if (p.bias) v = v + *reinterpret_cast<const f32x4*>(p.bias + col);
        *reinterpret_cast<f32x4*>(yr + col) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ dW, db
struct WgF32Params {
  const float* dy; int64_t ld_dy;
  const float* x;  int64_t ld_x;
  int64_t n_rows;
  int64_t n_steps;           // ceil(n_rows / 2): one MFMA k-step = two rows
  int64_t steps_per_block;
  float* partial;            // [gridDim.x][m_total * K + m_total]
  int m_off, m_total;        // this launch covers output rows m_off .. m_off + M of an [m_total, K] gradient
};


// The product reduces over the rows, so the MFMA's k index IS the row: lane l supplies dY[row0 + (l >> 5)][m0 + (l & 31)]
// and X[row0 + (l >> 5)][k0 + (l & 31)] -- 128-byte row segments straight from global memory, no transpose, no LDS.  As
// in the 16-bit kernel a persistent workgroup owns a slab of rows and keeps the whole [M, K] accumulator in registers
// (6 tiles of 32 x 32 per wave at most); db accumulates from the A operands on the side.
template <int M, int K, int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_f32_kernel(WgF32Params p) {
  constexpr int TM = M / 32, TK = K / 32;
  constexpr int WM = NW == 8 ? (TM % 4 == 0 ? 4 : 2) : 2, WK = NW / WM;
  static_assert(TM % WM == 0 && TK % WK == 0, "tile grid does not split over the waves");
  constexpr int MT = TM / WM, KT = TK / WK;
  static_assert(MT * KT <= 12, "accumulator does not fit the register file");
  // ring slots (steps of look-ahead + 1): few tiles per wave = few MFMAs per loaded operand = a memory-bound shape that
  // needs more loads in flight
  constexpr int R = MT * KT >= 6 ? 8 : (MT + KT <= 2 ? 24 : 12);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wk = wave / WM;
  const int r = lane & 31, h = lane >> 5;

  const int64_t s_beg = (int64_t)blockIdx.x * p.steps_per_block;
  int64_t s_end = s_beg + p.steps_per_block;
  if (s_end > p.n_steps) s_end = p.n_steps;

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

  const float* __restrict__ dyb = p.dy + 32 * wm * MT + r;
  const float* __restrict__ xb = p.x + 32 * wk * KT + r;
  // A ring of R steps (row pairs) in registers: a step's operands are requested R - 1 steps before its MFMAs and its
  // slot is refilled right after them, so R - 1 steps of loads are always in flight.  (With every CU streaming, a
  // 4-byte-per-lane load takes longer to come back than a few MFMAs last: with 4 steps of look-ahead the waves sat in
  // s_waitcnt 47 % of the time and the matrix pipe was 44 % busy -- PMC, round 3.)  Full steps -- both rows inside
  // the slab and the matrix -- run without any bounds arithmetic; the running read position stops at the last full
  // step (re-read, never out of bounds, values unused).
  const int64_t rows_end = 2 * s_end < p.n_rows ? 2 * s_end : p.n_rows;          // rows of this slab: [2 s_beg, rows_end)
  const int64_t full_steps = rows_end / 2 > s_beg ? rows_end / 2 - s_beg : 0;    // steps whose two rows both exist
  const int64_t ld_a = p.ld_dy, ld_b = p.ld_x;
  float fa[R][MT], fb[R][KT];
  const float* pa = dyb + (2 * s_beg + h) * ld_a;
  const float* pb = xb + (2 * s_beg + h) * ld_b;
  int64_t nxt = 0;                                     // step the next fetch reads (uniform)
  auto fetch = [&](int slot) {
#pragma unroll
    for (int a = 0; a < MT; ++a) fa[slot][a] = pa[32 * a];
#pragma unroll
    for (int b = 0; b < KT; ++b) fb[slot][b] = pb[32 * b];
    if (nxt + 1 < full_steps) { pa += 2 * ld_a; pb += 2 * ld_b; ++nxt; }
  };
  auto compute = [&](int slot) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      if (wk == 0) dbias[a] += fa[slot][a];
#pragma unroll
      for (int b = 0; b < KT; ++b) acc[a][b] = mfma_f32(fa[slot][a], fb[slot][b], acc[a][b]);
    }
  };
  if (full_steps > 0) {
#pragma unroll
    for (int j = 0; j < R; ++j) fetch(j);
    int64_t i = 0;
    for (; i + R <= full_steps; i += R) {
#pragma unroll
      for (int j = 0; j < R; ++j) { compute(j); fetch(j); }
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
      if (i + j < full_steps) compute(j);
  }
  // the rest of the slab: at the end of the matrix, one step with a single row
  for (int64_t s = s_beg + full_steps; s < s_end; ++s) {
    const int64_t row = 2 * s + h;
    const bool ok = row < rows_end;
    const int64_t rc = ok ? row : rows_end - 1;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const float v = dyb[rc * ld_a + 32 * a];
      const float av = ok ? v : 0.f;
      if (wk == 0) dbias[a] += av;
#pragma unroll
      for (int b = 0; b < KT; ++b) acc[a][b] = mfma_f32(av, xb[rc * ld_b + 32 * b], acc[a][b]);
    }
  }

  // acc tile (a, b) element e of lane l is dW[m][k]: m = 32 (wm MT + a) + (e & 3) + 8 (e >> 2) + 4 (l >> 5), k = 32 (wk KT + b) + (l & 31)
  float* out = p.partial + (int64_t)blockIdx.x * ((int64_t)p.m_total * K + p.m_total);
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = p.m_off + 32 * (wm * MT + a) + (e & 3) + 8 * (e >> 2) + 4 * h;
        out[m * K + 32 * (wk * KT + b) + r] = acc[a][b][e];
      }
  if (wk == 0) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const float d = dbias[a] + __shfl_xor(dbias[a], 32, 64);      // the two rows of every step
      if (h == 0) out[p.m_total * K + p.m_off + 32 * (wm * MT + a) + r] = d;
    }
  }
}

constexpr int f32_waves(int m, int k) {
  const int tiles = (m / 32) * (k / 32);
  return tiles >= 32 ? 8 : 4;          // (1024-thread workgroups would cap a wave at 128 registers)
}
constexpr int kNumCuF32 = 256;
constexpr int64_t kMinStepsPerBlock = 256;     // 512 rows per workgroup at least (cf. the 16-bit kernel)

// persistent workgroups: one per CU for the 8-wave shapes (their accumulators fill the register file at two waves per
// SIMD); the 4-wave shapes with <= 128 registers of accumulator + operands run two per CU (more loads in flight)
int64_t f32_grid(int64_t n_rows, int m, int k) {
  const int64_t steps = (n_rows + 1) / 2;
  const int64_t want = (steps + kMinStepsPerBlock - 1) / kMinStepsPerBlock;
  const int tiles_per_wave = (m / 32) * (k / 32) / f32_waves(m, k);
  const int64_t cap = (f32_waves(m, k) == 4 && tiles_per_wave <= 4) ? 2 * kNumCuF32 : kNumCuF32;
  return want < cap ? (want < 1 ? 1 : want) : cap;
}

template <int M, int K>
void launch_wgrad_f32(const WgF32Params& p, int64_t grid, hipStream_t stream) {
  constexpr int NW = f32_waves(M, K);
  hipLaunchKernelGGL((wgrad_f32_kernel<M, K, NW>), dim3((unsigned)grid), dim3(NW * 64), 0, stream, p);
}

}  // namespace

// (declared in common.h for csrc/linear.hip and csrc/linear_wgrad.hip, which dispatch on the dtype)
int linear_f32_launch(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t n_rows,
                      int k_in, int m_out, hipStream_t stream, const float* rowbias, const int32_t* rowidx, int64_t ld_rb,
                      const float* gate, int64_t ld_gate, int gate_kind, float* y_act, int64_t ld_yact, int act_kind) {
  LinF32Params p{static_cast<const float*>(x), ldx, static_cast<const float*>(w), bias, static_cast<float*>(y), ldy,
                 n_rows, m_out, rowbias, rowidx, ld_rb, gate, ld_gate, gate_kind, y_act, ld_yact, act_kind};
  if (y_act && k_in == 384) { set_error("segger_linear_fwd_f32_act: k_in 64, 128 or 256"); return SEGGER_EUNSUPPORTED; }
  if (gate && k_in == 384) { set_error("segger_linear_fwd_f32_gate: k_in 64, 128 or 256 on the exact kernel"); return SEGGER_EUNSUPPORTED; }
  if (rowbias && k_in == 384) { set_error("segger_linear_fwd_rowbias (fp32): k_in 64, 128 or 256"); return SEGGER_EUNSUPPORTED; }
  const int64_t nb = (n_rows + 127) / 128;
  if (nb > 0x7fffffffLL) { set_error("segger_linear_fwd: too many rows"); return SEGGER_EUNSUPPORTED; }
  dim3 grid((unsigned)nb), block(256);
  if (k_in == 384 && (m_out == 128 || m_out == 256)) {   // the data gradients of the stacked projections
    if (m_out == 128) hipLaunchKernelGGL((linear_f32_kslice_kernel<384, 128>), grid, block, 0, stream, p);
    else              hipLaunchKernelGGL((linear_f32_kslice_kernel<384, 256>), grid, block, 0, stream, p);
    SEGGER_LAUNCH_CHECK("linear_f32_kslice_kernel");
    return SEGGER_OK;
  }
  switch (k_in) {
    case 64:  hipLaunchKernelGGL((linear_f32_kernel<64>), grid, block, 0, stream, p); break;
    case 128: hipLaunchKernelGGL((linear_f32_kernel<128>), grid, block, 0, stream, p); break;
    case 256: hipLaunchKernelGGL((linear_f32_kernel<256>), grid, block, 0, stream, p); break;
    case 384: hipLaunchKernelGGL((linear_f32_kernel<384>), grid, block, 0, stream, p); break;
    default: set_error("segger_linear_fwd: k_in=%d not supported (64, 128, 256, 384)", k_in); return SEGGER_EUNSUPPORTED;
  }
  SEGGER_LAUNCH_CHECK("linear_f32_kernel");
  return SEGGER_OK;
}

size_t wgrad_f32_workspace_bytes(int64_t n_rows, int m_out, int k_in) {
  return (size_t)(f32_grid(n_rows, m_out, k_in) + kReduceGroups) * ((size_t)m_out * k_in + m_out) * sizeof(float);
}

int wgrad_f32_launch(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, int64_t n_rows, int m_out, int k_in,
                     float* partial, int64_t* n_slabs, hipStream_t stream) {
  const int64_t grid = f32_grid(n_rows, m_out, k_in);
  WgF32Params p{static_cast<const float*>(dy), ld_dy, static_cast<const float*>(x), ld_x, n_rows, (n_rows + 1) / 2, 0, partial,
                0, m_out};
  p.steps_per_block = (p.n_steps + grid - 1) / grid;
  *n_slabs = grid;
  if (m_out == 384 && k_in == 256) {
    // 96 accumulator tiles do not fit 8 waves at two waves per SIMD: two launches over the halves of dY's columns
    // (X is read twice; this shape is the first layer's concatenated input, once per step)
    for (int half = 0; half < 2; ++half) {
      WgF32Params q = p;
      q.dy = p.dy + 192 * half; q.m_off = 192 * half;
      launch_wgrad_f32<192, 256>(q, grid, stream);
    }
    SEGGER_LAUNCH_CHECK("wgrad_f32_kernel");
    return SEGGER_OK;
  }
#define CASE(MM, KK) if (m_out == MM && k_in == KK) { launch_wgrad_f32<MM, KK>(p, grid, stream); SEGGER_LAUNCH_CHECK("wgrad_f32_kernel"); return SEGGER_OK; }
  CASE(384, 128) CASE(384, 64)
  CASE(192, 256) CASE(192, 128) CASE(192, 64)
  CASE(128, 256) CASE(128, 128) CASE(128, 64)
  CASE(64, 256) CASE(64, 128) CASE(64, 64)
#undef CASE
  set_error("segger_linear_wgrad: m_out=%d k_in=%d not supported", m_out, k_in);
  return SEGGER_EUNSUPPORTED;
}

}  // namespace segger
