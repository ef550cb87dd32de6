// Host launchers for the fused GATv2 kernels (C ABI: include/segger_amd.h).
#include "gatv2_launch.h"
#include "draws.h"

namespace segger {

// grad_att / grad_bias = column sums of the slab [nblocks][width], in two deterministic stages:
// stage 1: kSlabSplits row ranges x 64-column chunks -> part[kSlabSplits][width]; stage 2: sum the splits.
constexpr int kSlabSplits = 128;

__global__ __launch_bounds__(256) void slab_reduce_stage1(const float* __restrict__ slab, int64_t nblocks, int width,
                                                         float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int colx = blockIdx.x * 64 + lane;
  const int64_t per = (nblocks + kSlabSplits - 1) / kSlabSplits;
  const int64_t r0 = (int64_t)blockIdx.y * per;
  const int64_t r1 = r0 + per < nblocks ? r0 + per : nblocks;
  float s = 0.f;
  if (colx < width)
    for (int64_t r = r0 + wave; r < r1; r += 4) s += slab[r * width + colx];
  sm[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && colx < width) part[(int64_t)blockIdx.y * width + colx] = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
}

__global__ __launch_bounds__(256) void slab_reduce_stage2(const float* __restrict__ part, int width, int hc,
                                                         float* __restrict__ grad_att, float* __restrict__ grad_bias) {
  const int colx = blockIdx.x * 256 + threadIdx.x;
  if (colx >= width) return;
  float t = 0.f;
  for (int r = 0; r < kSlabSplits; ++r) t += part[(int64_t)r * width + colx];
  if (colx < hc) grad_att[colx] = t;
  else if (grad_bias) grad_bias[colx - hc] = t;
}

// zero the first `pieces` 16-byte pieces of every row of a pitched matrix
__global__ __launch_bounds__(256) void zero_rows_kernel(char* __restrict__ base, int64_t pitch, int64_t pieces, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  *reinterpret_cast<u32x4*>(base + (i / pieces) * pitch + (i % pieces) * 16) = u32x4{0u, 0u, 0u, 0u};
}

__global__ __launch_bounds__(256) void dropout_bits_kernel(BitsParams p) { dropout_bits_body(p, blockIdx.x); }
__global__ __launch_bounds__(256) void dropout_bits_many_kernel(BitsJobs j) { dropout_bits_body(j.job[blockIdx.y], blockIdx.x); }

namespace {

// average-degree threshold above which the NG groups of a wave split ONE row
constexpr double kWavePerRowDegree = 32.0;

// set per call by the extern "C" entry points: generic kernels write grad_att / grad_bias themselves
struct GenericOut { float* grad_att = nullptr; float* grad_bias = nullptr; };

int launch(Pass pass, GatParams& p, int dtype, int heads, int channels, bool wpr, hipStream_t stream,
           const GenericOut& gen = GenericOut()) {
  if (!gatv2_has_specialised(heads, channels))
    return gatv2_launch_generic((int)pass, p, dtype, heads, channels, gen.grad_att, gen.grad_bias, stream);
  typedef int (*fn_t)(GatParams&, int, int, bool, hipStream_t);
  static const fn_t table[3][3] = {
      {gatv2_launch_fwd_f32, gatv2_launch_fwd_bf16, gatv2_launch_fwd_f16},
      {gatv2_launch_bwd_dst_f32, gatv2_launch_bwd_dst_bf16, gatv2_launch_bwd_dst_f16},
      {gatv2_launch_bwd_src_f32, gatv2_launch_bwd_src_bf16, gatv2_launch_bwd_src_f16}};
  if (dtype < 0 || dtype > 2) { set_error("gatv2: unknown dtype %d", dtype); return SEGGER_EINVAL; }
  return table[(int)pass][dtype](p, heads, channels, wpr, stream);
}

size_t elem_size(int dtype) { return dtype == SEGGER_F32 ? 4 : 2; }

// vectorised (specialised) kernels need 16-byte aligned rows; the generic kernels load element-wise
thread_local bool g_need_align = true;

int check_rows(const char* name, const void* ptr, int64_t ld, int dtype, int hc) {
  SEGGER_REQUIRE(ptr != nullptr, "gatv2: %s is NULL", name);
  SEGGER_REQUIRE(ld >= hc, "gatv2: ld of %s (%lld) < heads*channels (%d)", name, (long long)ld, hc);
  SEGGER_REQUIRE(ld * (int64_t)elem_size(dtype) < 0xffffffffLL, "gatv2: row stride of %s exceeds 4 GiB", name);
  if (g_need_align) {
    SEGGER_REQUIRE(aligned16(ptr), "gatv2: %s is not 16-byte aligned", name);
    SEGGER_REQUIRE((ld * (int64_t)elem_size(dtype)) % 16 == 0, "gatv2: row stride of %s is not a multiple of 16 bytes", name);
  }
  return SEGGER_OK;
}

int check_csr(const char* name, const segger_csr& g) {
  SEGGER_REQUIRE(g.n_rows >= 0 && g.n_cols >= 0 && g.n_edges >= 0, "gatv2: negative size in %s", name);
  SEGGER_REQUIRE(g.n_rows < 0x7fffffffLL && g.n_cols < 0x7fffffffLL && g.n_edges < 0x7fffffffLL,
                 "gatv2: %s exceeds 2^31-1 rows/cols/edges per batch", name);
  SEGGER_REQUIRE(g.indptr != nullptr, "gatv2: %s.indptr is NULL", name);
  SEGGER_REQUIRE(g.n_edges == 0 || g.col != nullptr, "gatv2: %s.col is NULL", name);
  return SEGGER_OK;
}

void set_dropout(GatParams& p, float dropout_p, uint64_t seed, const uint64_t* seed_dev) {
  p.drop_thr = 0; p.drop_scale = 1.f;
  if (dropout_p > 0.f) {
    p.drop_thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.drop_scale = 1.0f / (1.0f - dropout_p);
  }
  const uint64_t mixed = splitmix64(seed);
  p.seed_lo = (uint32_t)(mixed & 0xffffffffu);
  p.seed_hi = (uint32_t)(mixed >> 32);
  p.seed_raw = seed;
  p.seed_dev = seed_dev;
}

bool use_wave_per_row(const segger_csr& g) {
  return g.n_rows > 0 && (double)g.n_edges / (double)g.n_rows >= kWavePerRowDegree;
}

}  // namespace
}  // namespace segger

using namespace segger;

#define CHECK_RC(expr) do { int _rc = (expr); if (_rc != SEGGER_OK) return _rc; } while (0)

// argument checks + kernel parameters of one forward; *empty = nothing to launch
static int fwd_params(const segger_gatv2_fwd_args* a, GatParams& p, bool* empty) {
  *empty = false;
  SEGGER_REQUIRE(a != nullptr, "segger_gatv2_fwd: args is NULL");
  SEGGER_REQUIRE(a->heads > 0 && a->channels > 0, "segger_gatv2_fwd: heads/channels must be positive");
  const int hc = a->heads * a->channels;
  g_need_align = gatv2_has_specialised(a->heads, a->channels);
  CHECK_RC(check_csr("by_dst", a->by_dst));
  SEGGER_REQUIRE(a->att != nullptr, "segger_gatv2_fwd: att is NULL");
  SEGGER_REQUIRE(a->negative_slope >= 0.f && a->negative_slope <= 1.f, "segger_gatv2_fwd: negative_slope must be in [0,1]");
  SEGGER_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "segger_gatv2_fwd: dropout_p must be in [0,1)");
  if (a->by_dst.n_rows == 0) { *empty = true; return SEGGER_OK; }
  CHECK_RC(check_rows("x_r", a->x_r, a->ld_xr, a->dtype, hc));
  CHECK_RC(check_rows("out", a->out, a->ld_out, a->dtype, hc));
  if (a->by_dst.n_edges > 0) CHECK_RC(check_rows("x_l", a->x_l, a->ld_xl, a->dtype, hc));
  if (a->pre) {
    CHECK_RC(check_rows("pre", a->pre, a->ld_pre, a->dtype, hc));
    SEGGER_REQUIRE(!(a->pre == a->out && a->apply_gelu), "segger_gatv2_fwd: pre may alias out only without GELU");
  }
  SEGGER_REQUIRE(!((a->dropout_p > 0.f && !a->keep_bits) || a->alpha) || a->by_dst.n_edges == 0 || a->by_dst.eid != nullptr,
                 "segger_gatv2_fwd: by_dst.eid is required for dropout (without keep_bits) / alpha output");
  p = GatParams{};
  p.indptr = a->by_dst.indptr; p.col = a->by_dst.col; p.eid = a->by_dst.eid; p.order = a->by_dst.row_order;
  p.n_rows = a->by_dst.n_rows; p.n_edges = a->by_dst.n_edges;
  if (a->by_dst.blk_cnt && a->by_dst.blk_src && a->by_dst.col_local) {
    p.blk_cnt = a->by_dst.blk_cnt; p.blk_src = a->by_dst.blk_src; p.col_local = a->by_dst.col_local;
  }
  p.xl = a->x_l; p.ld_xl = a->ld_xl; p.xr = a->x_r; p.ld_xr = a->ld_xr;
  p.att = a->att; p.bias = a->bias;
  p.out = a->out; p.ld_out = a->ld_out; p.pre = a->pre; p.ld_pre = a->ld_pre;
  p.lse = a->lse; p.alpha = a->alpha;
  p.bits = a->alpha ? nullptr : a->keep_bits;
  p.slope = a->negative_slope; p.apply_gelu = a->apply_gelu; p.rows_per_wave_iter = 1;
  set_dropout(p, a->dropout_p, a->seed, a->seed_dev);
  return SEGGER_OK;
}

extern "C" int segger_gatv2_fwd(const segger_gatv2_fwd_args* a, segger_stream_t stream) {
  GatParams p;
  bool empty = false;
  CHECK_RC(fwd_params(a, p, &empty));
  if (empty) return SEGGER_OK;
  return launch(Pass::Fwd, p, a->dtype, a->heads, a->channels, use_wave_per_row(a->by_dst), (hipStream_t)stream);
}

extern "C" int segger_gatv2_fwd_pair(const segger_gatv2_fwd_args* a, const segger_gatv2_fwd_args* b, segger_stream_t stream) {
  GatParams pa, pb;
  bool ea = false, eb = false;
  CHECK_RC(fwd_params(a, pa, &ea));
  CHECK_RC(fwd_params(b, pb, &eb));
  // one launch when `a` is a low-degree (group-per-row) and `b` a high-degree (wave-per-row) edge type of the same
  // specialised geometry and storage type and neither asks for attention weights; otherwise two launches
  const bool one = !ea && !eb && a->dtype == b->dtype && a->heads == b->heads && a->channels == b->channels &&
                   gatv2_has_specialised(a->heads, a->channels) && !use_wave_per_row(a->by_dst) &&
                   use_wave_per_row(b->by_dst) && !a->alpha && !b->alpha;
  if (!one) {
    if (!ea) CHECK_RC(launch(Pass::Fwd, pa, a->dtype, a->heads, a->channels, use_wave_per_row(a->by_dst), (hipStream_t)stream));
    if (!eb) CHECK_RC(launch(Pass::Fwd, pb, b->dtype, b->heads, b->channels, use_wave_per_row(b->by_dst), (hipStream_t)stream));
    return SEGGER_OK;
  }
  switch (a->dtype) {
    case SEGGER_F32:  return gatv2_launch_fwd_pair_f32(pa, pb, a->heads, a->channels, (hipStream_t)stream);
    case SEGGER_BF16: return gatv2_launch_fwd_pair_bf16(pa, pb, a->heads, a->channels, (hipStream_t)stream);
    case SEGGER_F16:  return gatv2_launch_fwd_pair_f16(pa, pb, a->heads, a->channels, (hipStream_t)stream);
  }
  set_error("gatv2: unknown dtype %d", a->dtype);
  return SEGGER_EINVAL;
}

extern "C" size_t segger_gatv2_bwd_workspace_bytes(int64_t n_dst, int32_t heads, int32_t channels) {
  if (n_dst <= 0 || heads <= 0 || channels <= 0) return 16;
  // upper bound over both modes: wave-per-row has the most blocks (4 rows per block-iteration)
  const int64_t per = 4 * (int64_t)bwd_row_iters(n_dst);
  const int64_t blocks = (n_dst + per - 1) / per + kNumXcd;
  return (size_t)(blocks + kSlabSplits) * 2 * heads * channels * sizeof(float);
}

// ---- backward in phases (shared by segger_gatv2_bwd and segger_gatv2_bwd_pair) ------------------------------
namespace {
struct BwdState {
  GatParams p;
  bool direct = false, specialised = false;
  int64_t n_dst = 0, n_src = 0, n_edges = 0;
};

// argument checks + the parameters both passes share
int bwd_prepare(const segger_gatv2_bwd_args* a, BwdState& s) {
  SEGGER_REQUIRE(a != nullptr, "segger_gatv2_bwd: args is NULL");
  SEGGER_REQUIRE(a->heads > 0 && a->channels > 0, "segger_gatv2_bwd: heads/channels must be positive");
  const int hc = a->heads * a->channels;
  const bool specialised = gatv2_has_specialised(a->heads, a->channels);
  g_need_align = specialised;
  CHECK_RC(check_csr("by_dst", a->by_dst));
  const bool direct = a->src_unique != 0;
  SEGGER_REQUIRE(!direct || specialised, "segger_gatv2_bwd: src_unique needs a specialised (heads, channels) geometry");
  if (!direct) {
    CHECK_RC(check_csr("by_src", a->by_src));
    SEGGER_REQUIRE(a->by_dst.n_edges == a->by_src.n_edges && a->by_dst.n_rows == a->by_src.n_cols &&
                   a->by_dst.n_cols == a->by_src.n_rows, "segger_gatv2_bwd: by_dst and by_src describe different graphs");
  }
  SEGGER_REQUIRE(a->att && a->grad_att, "segger_gatv2_bwd: att / grad_att is NULL");
  SEGGER_REQUIRE(a->negative_slope >= 0.f && a->negative_slope <= 1.f, "segger_gatv2_bwd: negative_slope must be in [0,1]");
  SEGGER_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, "segger_gatv2_bwd: dropout_p must be in [0,1)");
  const int64_t n_dst = a->by_dst.n_rows, n_src = a->by_dst.n_cols, n_edges = a->by_dst.n_edges;
  if (n_dst > 0) {
    CHECK_RC(check_rows("x_r", a->x_r, a->ld_xr, a->dtype, hc));
    CHECK_RC(check_rows("grad_out", a->grad_out, a->ld_go, a->dtype, hc));
    CHECK_RC(check_rows("pre", a->pre, a->ld_pre, a->dtype, hc));
    CHECK_RC(check_rows("grad_pre", a->grad_pre, a->ld_gp, a->dtype, hc));
    CHECK_RC(check_rows("grad_xr", a->grad_xr, a->ld_gxr, a->dtype, hc));
    SEGGER_REQUIRE(a->lse && a->dsum, "segger_gatv2_bwd: lse / dsum is NULL");
  }
  if (n_src > 0) {
    CHECK_RC(check_rows("x_l", a->x_l, a->ld_xl, a->dtype, hc));
    CHECK_RC(check_rows("grad_xl", a->grad_xl, a->ld_gxl, a->dtype, hc));
  }
  SEGGER_REQUIRE(!(a->dropout_p > 0.f) || n_edges == 0 ||
                     ((a->by_dst.eid || a->keep_bits_dst) && (direct || a->by_src.eid || a->keep_bits_src)),
                 "segger_gatv2_bwd: eid arrays (or keep_bits) are required for dropout");
  const size_t need = segger_gatv2_bwd_workspace_bytes(n_dst, a->heads, a->channels);
  if (a->workspace == nullptr || a->workspace_bytes < need) {
    set_error("segger_gatv2_bwd: workspace %zu < %zu bytes", a->workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  SEGGER_REQUIRE(!a->zero_rows_out || (!direct && specialised),
                 "segger_gatv2_bwd: zero_rows_out needs the two-pass backward of a specialised geometry");
  if (a->zero_rows_out && n_src > 0) CHECK_RC(check_rows("zero_rows_out", a->zero_rows_out, a->ld_zero, a->dtype, hc));

  s.direct = direct; s.specialised = specialised; s.n_dst = n_dst; s.n_src = n_src; s.n_edges = n_edges;
  GatParams& p = s.p;
  p = GatParams{};
  p.xl = a->x_l; p.ld_xl = a->ld_xl; p.xr = a->x_r; p.ld_xr = a->ld_xr;
  p.att = a->att; p.bias = a->bias;
  p.pre = const_cast<void*>(a->pre); p.ld_pre = a->ld_pre; p.lse = const_cast<float*>(a->lse);
  p.gout = a->grad_out; p.ld_go = a->ld_go; p.gpre = a->grad_pre; p.ld_gp = a->ld_gp; p.dsum = a->dsum;
  p.gxl = a->grad_xl; p.ld_gxl = a->ld_gxl; p.gxr = a->grad_xr; p.ld_gxr = a->ld_gxr;
  p.slab = static_cast<float*>(a->workspace);
  p.slope = a->negative_slope; p.apply_gelu = a->apply_gelu;
  set_dropout(p, a->dropout_p, a->seed, a->seed_dev);
  return SEGGER_OK;
}

// parameters of the destination-side pass (rows = destinations)
void bwd_dst_params(const segger_gatv2_bwd_args* a, BwdState& s) {
  GatParams& p = s.p;
  p.indptr = a->by_dst.indptr; p.col = a->by_dst.col; p.eid = a->by_dst.eid; p.order = a->by_dst.row_order;
  p.n_rows = s.n_dst; p.n_edges = s.n_edges; p.rows_per_wave_iter = bwd_row_iters(s.n_dst);
  p.bits = a->keep_bits_dst;
  p.direct_gxl = s.direct ? 1 : 0;
  p.zero_rows = nullptr; p.ld_zero = 0;
}

// parameters of the source-side pass (rows = sources)
void bwd_src_params(const segger_gatv2_bwd_args* a, BwdState& s) {
  GatParams& p = s.p;
  p.direct_gxl = 0;
  p.indptr = a->by_src.indptr; p.col = a->by_src.col; p.eid = a->by_src.eid; p.order = a->by_src.row_order;
  p.bits = a->keep_bits_src;
  p.n_rows = s.n_src; p.rows_per_wave_iter = 1;
  p.zero_rows = a->zero_rows_out; p.ld_zero = a->ld_zero;
}

// the one-pass form's own zero fill: sources without an out-edge keep a zero gradient, the others are stored by the
// destination pass (rows are 16-byte aligned multiples of 16 bytes: checked above.  hipMemset2DAsync measured 0.19 ms
// for 1M x 256 B at pitch 768; this kernel 0.05 ms)
int bwd_zero_fill(const segger_gatv2_bwd_args* a, const BwdState& s, hipStream_t stream) {
  const size_t es = elem_size(a->dtype);
  const int64_t pieces = (int64_t)a->heads * a->channels * es / 16, total = s.n_src * pieces;
  hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     static_cast<char*>(a->grad_xl), (int64_t)a->ld_gxl * (int64_t)es, pieces, total);
  SEGGER_LAUNCH_CHECK("zero_rows_kernel");
  return SEGGER_OK;
}

// grad_att / grad_bias from the per-block slabs a destination pass (s.p still holds its block count) left behind
int bwd_reduce_slab(const segger_gatv2_bwd_args* a, const BwdState& s, hipStream_t stream) {
  if (!s.specialised) return SEGGER_OK;                   // the generic kernels write grad_att / grad_bias themselves
  const GatParams& p = s.p;
  const int hc = a->heads * a->channels, width = 2 * hc;
  float* part = p.slab + p.nblocks * width;              // behind the per-block slabs
  if (defer_reduce(ReduceSeg{p.slab, p.nblocks, width, hc, a->grad_att, a->grad_bias, part}, stream)) return SEGGER_OK;
  hipLaunchKernelGGL(slab_reduce_stage1, dim3((width + 63) / 64, kSlabSplits), dim3(256), 0, stream,
                     p.slab, p.nblocks, width, part);
  hipLaunchKernelGGL(slab_reduce_stage2, dim3((width + 255) / 256), dim3(256), 0, stream,
                     part, width, hc, a->grad_att, a->grad_bias);
  SEGGER_LAUNCH_CHECK("slab_reduce kernels");
  return SEGGER_OK;
}

int bwd_no_destinations(const segger_gatv2_bwd_args* a, hipStream_t stream) {
  const int hc = a->heads * a->channels;
  SEGGER_HIP(hipMemsetAsync(a->grad_att, 0, hc * sizeof(float), stream));
  if (a->grad_bias) SEGGER_HIP(hipMemsetAsync(a->grad_bias, 0, hc * sizeof(float), stream));
  return SEGGER_OK;
}

// largest tx-neighbors-tx source count for which segger_gatv2_bwd_pair merges the launches (what it saves is a fixed
// ~5 us per layer: the launch of a kernel that is latency-bound at these sizes)
int64_t bwd_pair_max_rows() {
  static const int64_t v = [] {
    const char* e = getenv("SEGGER_BWD_PAIR_MAX_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)SEGGER_BWD_PAIR_MAX_ROWS;
  }();
  return v;
}
}  // namespace

extern "C" int segger_gatv2_bwd(const segger_gatv2_bwd_args* a, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BwdState s;
  CHECK_RC(bwd_prepare(a, s));
  SEGGER_REQUIRE(a->passes >= 0 && a->passes <= 3, "segger_gatv2_bwd: passes must be 0 (both), 1 (destination), 2 (source) or 3");
  // ---- destination side ------------------------------------------------------
  if (a->passes != 2) {
    bwd_dst_params(a, s);
    if (s.direct && s.n_src > 0 && !a->grad_xl_zeroed) CHECK_RC(bwd_zero_fill(a, s, stream));
    if (s.n_dst > 0) {
      GenericOut gen; gen.grad_att = a->grad_att; gen.grad_bias = a->grad_bias;
      CHECK_RC(launch(Pass::BwdDst, s.p, a->dtype, a->heads, a->channels, use_wave_per_row(a->by_dst), stream, gen));
      if (a->passes != 3) CHECK_RC(bwd_reduce_slab(a, s, stream));
    } else {
      CHECK_RC(bwd_no_destinations(a, stream));
    }
  }
  // ---- source side -----------------------------------------------------------
  if (s.direct || a->passes == 1 || a->passes == 3) return SEGGER_OK;
  bwd_src_params(a, s);
  if (s.n_src > 0) CHECK_RC(launch(Pass::BwdSrc, s.p, a->dtype, a->heads, a->channels, use_wave_per_row(a->by_src), stream));
  return SEGGER_OK;
}

extern "C" int segger_gatv2_bwd_pair(const segger_gatv2_bwd_args* a, const segger_gatv2_bwd_args* b, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BwdState sa, sb;
  CHECK_RC(bwd_prepare(a, sa));
  CHECK_RC(bwd_prepare(b, sb));
  // one launch for a's source pass and b's one-pass destination pass when: a is a two-pass, group-per-row edge type
  // and b a one-pass, wave-per-row one of the same specialised geometry and storage type; both non-empty; b's grad_xl
  // is the matrix a was asked to zero-fill (or a was asked for none and b's is not zeroed yet) and a's destinations
  // index the same nodes as b's sources -- a's DESTINATION pass then zero-fills it, since its source pass now runs
  // beside b's stores; and the batch is small enough for the merger to pay (bwd_pair_max_rows)
  const bool zero_by_a = a->zero_rows_out ? (a->zero_rows_out == b->grad_xl && a->ld_zero == b->ld_gxl && b->grad_xl_zeroed)
                                          : !b->grad_xl_zeroed;
  const bool one = !sa.direct && sb.direct && a->dtype == b->dtype && a->heads == b->heads && a->channels == b->channels &&
                   sa.specialised && sa.n_dst > 0 && sa.n_src > 0 && sb.n_dst > 0 && sb.n_src > 0 && sa.n_dst == sb.n_src &&
                   !use_wave_per_row(a->by_dst) && !use_wave_per_row(a->by_src) && use_wave_per_row(b->by_dst) &&
                   zero_by_a && sa.n_src <= bwd_pair_max_rows();
  if (!one) {
    CHECK_RC(segger_gatv2_bwd(a, stream_));
    return segger_gatv2_bwd(b, stream_);
  }
  bwd_dst_params(a, sa);
  sa.p.zero_rows = b->grad_xl; sa.p.ld_zero = b->ld_gxl;
  CHECK_RC(launch(Pass::BwdDst, sa.p, a->dtype, a->heads, a->channels, false, stream));
  CHECK_RC(bwd_reduce_slab(a, sa, stream));
  bwd_src_params(a, sa);
  sa.p.zero_rows = nullptr; sa.p.ld_zero = 0;
  bwd_dst_params(b, sb);
  switch (a->dtype) {
    case SEGGER_F32:  CHECK_RC(gatv2_launch_bwd_src_dst_pair_f32(sa.p, sb.p, a->heads, a->channels, stream)); break;
    case SEGGER_BF16: CHECK_RC(gatv2_launch_bwd_src_dst_pair_bf16(sa.p, sb.p, a->heads, a->channels, stream)); break;
    case SEGGER_F16:  CHECK_RC(gatv2_launch_bwd_src_dst_pair_f16(sa.p, sb.p, a->heads, a->channels, stream)); break;
    default: set_error("gatv2: unknown dtype %d", a->dtype); return SEGGER_EINVAL;
  }
  return bwd_reduce_slab(b, sb, stream);
}

extern "C" int segger_gatv2_has_specialised(int32_t heads, int32_t channels) {
  return gatv2_has_specialised(heads, channels) ? 1 : 0;
}

extern "C" int segger_dropout_bits(const int32_t* eid, int64_t n_edges, int32_t heads, float dropout_p,
                                   const uint64_t* seeds, int32_t n_seeds, const uint64_t* seed_dev, uint8_t* bits,
                                   int64_t plane_stride, segger_stream_t stream) {
  SEGGER_REQUIRE(n_edges >= 0 && heads > 0 && heads <= 8, "segger_dropout_bits: heads must be in 1..8");
  SEGGER_REQUIRE(n_seeds > 0 && n_seeds <= 16 && seeds, "segger_dropout_bits: 1..16 seeds");
  SEGGER_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "segger_dropout_bits: dropout_p must be in [0,1)");
  if (n_edges == 0) return SEGGER_OK;
  SEGGER_REQUIRE(eid && bits, "segger_dropout_bits: NULL pointer");
  SEGGER_REQUIRE(plane_stride >= n_edges && plane_stride % 4 == 0 && ((uintptr_t)bits & 3u) == 0,
                 "segger_dropout_bits: plane_stride must be a multiple of 4 >= n_edges and bits 4-byte aligned");
  BitsParams p{};
  p.eid = eid; p.n_edges = n_edges; p.plane_stride = plane_stride; p.heads = heads; p.n_seeds = n_seeds;
  p.seed_dev = seed_dev; p.bits = bits; p.eid_aligned = aligned16(eid) ? 1 : 0;
  p.thr = (uint32_t)((double)dropout_p * 16777216.0);
  for (int l = 0; l < n_seeds; ++l) p.seeds[l] = seeds[l];
  hipLaunchKernelGGL(dropout_bits_kernel, dim3((unsigned)((n_edges + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, p);
  SEGGER_LAUNCH_CHECK("dropout_bits_kernel");
  return SEGGER_OK;
}

namespace segger {
int fill_bits_params(const segger_bits_job& jb, int heads, float dropout_p, const uint64_t* seed_dev, int i, BitsParams* out) {
  SEGGER_REQUIRE(jb.n_edges >= 0 && jb.n_seeds > 0 && jb.n_seeds <= 16, "segger_dropout_bits_many: job %d: bad sizes", i);
  SEGGER_REQUIRE(jb.n_edges == 0 || (jb.eid && jb.bits), "segger_dropout_bits_many: job %d: NULL pointer", i);
  SEGGER_REQUIRE(jb.plane_stride >= jb.n_edges && jb.plane_stride % 4 == 0 && ((uintptr_t)jb.bits & 3u) == 0,
                 "segger_dropout_bits_many: job %d: plane_stride must be a multiple of 4 >= n_edges, bits 4-byte aligned", i);
  BitsParams& p = *out;
  p = BitsParams{};
  p.eid = jb.eid; p.n_edges = jb.n_edges; p.plane_stride = jb.plane_stride; p.heads = heads; p.n_seeds = jb.n_seeds;
  p.seed_dev = seed_dev; p.bits = jb.bits; p.eid_aligned = aligned16(jb.eid) ? 1 : 0;
  p.thr = (uint32_t)((double)dropout_p * 16777216.0);
  for (int l = 0; l < jb.n_seeds; ++l) p.seeds[l] = jb.seeds[l];
  return SEGGER_OK;
}
}  // namespace segger

extern "C" int segger_dropout_bits_many(const segger_bits_job* jobs, int32_t n_jobs, int32_t heads, float dropout_p,
                                        const uint64_t* seed_dev, segger_stream_t stream) {
  SEGGER_REQUIRE(jobs && n_jobs > 0 && n_jobs <= kMaxBitsJobs, "segger_dropout_bits_many: 1..4 jobs");
  SEGGER_REQUIRE(heads > 0 && heads <= 8, "segger_dropout_bits_many: heads must be in 1..8");
  SEGGER_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "segger_dropout_bits_many: dropout_p must be in [0,1)");
  BitsJobs all{};
  int64_t most = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const int rc = fill_bits_params(jobs[i], heads, dropout_p, seed_dev, i, &all.job[i]);
    if (rc != SEGGER_OK) return rc;
    if (jobs[i].n_edges > most) most = jobs[i].n_edges;
  }
  if (most == 0) return SEGGER_OK;
  hipLaunchKernelGGL(dropout_bits_many_kernel, dim3((unsigned)((most + 1023) / 1024), (unsigned)n_jobs), dim3(256), 0,
                     (hipStream_t)stream, all);
  SEGGER_LAUNCH_CHECK("dropout_bits_many_kernel");
  return SEGGER_OK;
}

namespace segger {
namespace {
__global__ void step_advance_kernel(int64_t* step, int64_t inc, int64_t* copy) {
  if (threadIdx.x == 0) { const int64_t v = *step + inc; *step = v; if (copy) *copy = v; }
}
}  // namespace
}  // namespace segger

extern "C" int segger_step_advance(int64_t* step, int64_t inc, int64_t* copy, segger_stream_t stream) {
  SEGGER_REQUIRE(step != nullptr, "segger_step_advance: step is NULL");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, inc, copy);
  SEGGER_LAUNCH_CHECK("step_advance_kernel");
  return SEGGER_OK;
}
