// Per-(pass, dtype) launch entry points.  Each is instantiated in its own
// translation unit (gatv2_inst.hip compiled with -DSEGGER_INST_PASS/-DSEGGER_INST_DTYPE)
// so the build parallelises; gatv2.hip holds the C ABI and argument checks.
#pragma once
#include "gatv2_kernels.h"

namespace segger {

enum class Pass : int { Fwd = 0, BwdDst = 1, BwdSrc = 2 };

// destination rows walked per wave in the dst-side backward (amortises the
// per-block grad_att / grad_bias partial slab)
constexpr int kBwdRowIters = 8;
// ... but small graphs (tile batches of ~50 k nodes) keep one row batch per wave so the grid still has
// thousands of blocks: iterations grow with the row count, 1 below 32 k rows, 8 from 256 k rows up
static inline int bwd_row_iters(int64_t n_rows) {
  const int64_t it = n_rows / 32768;
  return (int)(it < 1 ? 1 : (it > kBwdRowIters ? kBwdRowIters : it));
}

// segger_gatv2_bwd_pair merges launches up to this many source rows of the two-pass edge type
#ifndef SEGGER_BWD_PAIR_MAX_ROWS
#define SEGGER_BWD_PAIR_MAX_ROWS 262144
#endif

// (heads, channels/8) combinations with a specialised kernel
#define SEGGER_GEOMETRIES(X) \
  X(1, 4) X(2, 4) X(3, 4) X(4, 4) \
  X(1, 8) X(2, 8) X(3, 8) X(4, 8)

#define SEGGER_DECL_LAUNCH(name) \
  int name(GatParams& p, int heads, int channels, bool wave_per_row, hipStream_t stream);
SEGGER_DECL_LAUNCH(gatv2_launch_fwd_f32)
SEGGER_DECL_LAUNCH(gatv2_launch_fwd_bf16)
SEGGER_DECL_LAUNCH(gatv2_launch_fwd_f16)
SEGGER_DECL_LAUNCH(gatv2_launch_bwd_dst_f32)
SEGGER_DECL_LAUNCH(gatv2_launch_bwd_dst_bf16)
SEGGER_DECL_LAUNCH(gatv2_launch_bwd_dst_f16)
SEGGER_DECL_LAUNCH(gatv2_launch_bwd_src_f32)
SEGGER_DECL_LAUNCH(gatv2_launch_bwd_src_bf16)
SEGGER_DECL_LAUNCH(gatv2_launch_bwd_src_f16)
#undef SEGGER_DECL_LAUNCH
// forward of two edge types in one launch: `a` group-per-row, `b` wave-per-row (gatv2_fwd_pair_kernel)
int gatv2_launch_fwd_pair_f32(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream);
int gatv2_launch_fwd_pair_bf16(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream);
int gatv2_launch_fwd_pair_f16(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream);
// source pass of `a` (group-per-row) + one-pass destination pass of `b` (wave-per-row, DIRECT) in one launch
int gatv2_launch_bwd_src_dst_pair_f32(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream);
int gatv2_launch_bwd_src_dst_pair_bf16(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream);
int gatv2_launch_bwd_src_dst_pair_f16(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream);

// any other (heads, channels <= 512): one wave per row, element loads (gatv2_generic.hip)
bool gatv2_has_specialised(int heads, int channels);
int gatv2_launch_generic(int pass, GatParams& p, int dtype, int heads, int channels, float* grad_att, float* grad_bias,
                         hipStream_t stream);

}  // namespace segger
