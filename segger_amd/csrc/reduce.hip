// All final sums of a step's per-workgroup partials in ONE launch.
//
// The weight-gradient kernels (csrc/linear_wgrad.hip) and the destination pass of the GATv2 backward (csrc/gatv2.hip)
// leave [n_slabs][width] fp32 partials that a small kernel sums in slab order.  At segger's default batch budget
// (1M edges: ~50k transcripts) a training step holds ~30 such sums of 4-8 us each -- a quarter of the kernel nodes of
// the captured step (segger_amd/train_step_graph.py) for 2 % of its arithmetic.  A caller that can wait for the sums
// until the end of the backward pass brackets it with segger_reductions_defer_begin() / segger_reductions_flush():
// the producers then only queue their sum (a process-wide table, passed to the kernel by value: nothing to allocate,
// capturable) and the flush runs them all as one grid.  Process-wide, not thread-local: torch's autograd engine runs
// the backward nodes that queue the sums on its own device thread, not on the thread that brackets the pass; one
// process drives one GPU stream here, so a mutex around the table is all the protection it needs.  Deterministic: every column is summed by one thread in slab
// order, exactly as the per-producer kernels do.
#include "common.h"
#include <mutex>

namespace segger {
namespace {

constexpr int kMaxSegs = 56;                    // 56 x 56 B + 8 by value in the kernel arguments (< 4 KiB)
constexpr int64_t kOnePassSlabs = 128;          // longer sums first fold into kReduceGroups interleaved groups

struct ReduceBatch { int n; ReduceSeg seg[kMaxSegs]; };
static ReduceBatch g_batch;
static bool g_active = false;
static int g_device = -1;                       // the device that opened the bracket
static hipStream_t g_stream = nullptr;          // the stream of the first producer that queued a sum
static bool g_stream_set = false;
static std::mutex g_mu;

__global__ __launch_bounds__(256) void reduce_many_kernel(ReduceBatch b) {
  const ReduceSeg g = b.seg[blockIdx.y];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= g.width) return;
  const float* __restrict__ p = g.partial + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t s = 0;
  for (; s + 3 < g.n_slabs; s += 4) {
    s0 += p[s * g.width];       s1 += p[(s + 1) * g.width];
    s2 += p[(s + 2) * g.width]; s3 += p[(s + 3) * g.width];
  }
  for (; s < g.n_slabs; ++s) s0 += p[s * g.width];
  const float t = (s0 + s1) + (s2 + s3);
  if (e < g.split) g.out0[e] = t;
  else if (g.out1) g.out1[e - g.split] = t;
}

// long sums, first level: scratch[g][e] = sum of the slabs g, g + G, g + 2 G, ... (grid: columns x segments x groups)
__global__ __launch_bounds__(256) void reduce_many_fold_kernel(ReduceBatch b) {
  const ReduceSeg g = b.seg[blockIdx.y];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= g.width) return;
  const float* __restrict__ p = g.partial + e;
  const int grp = blockIdx.z;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t s = grp;
  for (; s + 3 * kReduceGroups < g.n_slabs; s += 4 * kReduceGroups) {
    s0 += p[s * g.width];                       s1 += p[(s + kReduceGroups) * g.width];
    s2 += p[(s + 2 * kReduceGroups) * g.width]; s3 += p[(s + 3 * kReduceGroups) * g.width];
  }
  for (; s < g.n_slabs; s += kReduceGroups) s0 += p[s * g.width];
  g.scratch[(int64_t)grp * g.width + e] = (s0 + s1) + (s2 + s3);
}

}  // namespace

bool defer_reduce(const ReduceSeg& seg, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_active || g_batch.n >= kMaxSegs || (seg.n_slabs > kOnePassSlabs && seg.scratch == nullptr)) return false;
  // the table belongs to ONE device and ONE stream (the bracket's): a backward of another model on another GPU or
  // stream of this process must not leave its sum to a flush that runs elsewhere -- it launches its own
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != g_device) return false;
  if (!g_stream_set) { g_stream = stream; g_stream_set = true; }
  else if (stream != g_stream) return false;
  g_batch.seg[g_batch.n++] = seg;
  return true;
}

}  // namespace segger

using namespace segger;

extern "C" int segger_reductions_defer_begin(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  SEGGER_REQUIRE(!g_active, "segger_reductions_defer_begin: already deferring (flush first)");
  SEGGER_HIP(hipGetDevice(&g_device));
  g_active = true;
  g_stream_set = false;
  g_batch.n = 0;
  return SEGGER_OK;
}

extern "C" int segger_reductions_pending(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_active ? g_batch.n : -1;
}

extern "C" int segger_reductions_flush(segger_stream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  SEGGER_REQUIRE(g_active, "segger_reductions_flush: nothing is being deferred");
  g_active = false;
  if (g_batch.n == 0) return SEGGER_OK;
  if (g_stream_set && g_stream != (hipStream_t)stream) {
    g_batch.n = 0;
    set_error("segger_reductions_flush: the queued sums were produced on another stream than the one given");
    return SEGGER_EINVAL;
  }
  // sums of more than kOnePassSlabs partials: one launch folds them all into kReduceGroups groups each (their scratch),
  // and the final launch sums those groups like any short sum
  ReduceBatch longs; longs.n = 0;
  int64_t long_width = 0;
  for (int i = 0; i < g_batch.n; ++i) {
    ReduceSeg& g = g_batch.seg[i];
    if (g.n_slabs <= kOnePassSlabs) continue;
    longs.seg[longs.n++] = g;
    long_width = g.width > long_width ? g.width : long_width;
    g.partial = g.scratch; g.n_slabs = kReduceGroups;
  }
  if (longs.n > 0) {
    hipLaunchKernelGGL(reduce_many_fold_kernel, dim3((unsigned)((long_width + 255) / 256), (unsigned)longs.n, kReduceGroups),
                       dim3(256), 0, (hipStream_t)stream, longs);
    if (hipGetLastError() != hipSuccess) { g_batch.n = 0; set_error("reduce_many_fold_kernel: launch failed"); return SEGGER_EHIP; }
  }
  int64_t width = 0;
  for (int i = 0; i < g_batch.n; ++i) width = g_batch.seg[i].width > width ? g_batch.seg[i].width : width;
  hipLaunchKernelGGL(reduce_many_kernel, dim3((unsigned)((width + 255) / 256), (unsigned)g_batch.n), dim3(256), 0,
                     (hipStream_t)stream, g_batch);
  g_batch.n = 0;
  SEGGER_LAUNCH_CHECK("reduce_many_kernel");
  return SEGGER_OK;
}
