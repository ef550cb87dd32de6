// One (pass, dtype) slice of the fused GATv2 kernels.  Build with
//   -DSEGGER_INST_PASS={0 fwd,1 bwd_dst,2 bwd_src} -DSEGGER_INST_DTYPE={0 f32,1 bf16,2 f16}
#include "gatv2_launch.h"

#if SEGGER_INST_DTYPE == 0
#define INST_T float
#define INST_TN f32
#elif SEGGER_INST_DTYPE == 1
#define INST_T bf16_t
#define INST_TN bf16
#else
#define INST_T f16_t
#define INST_TN f16
#endif
#if SEGGER_INST_PASS == 0
#define INST_KERNEL gatv2_fwd_kernel
#define INST_PN fwd
#elif SEGGER_INST_PASS == 1
#define INST_KERNEL gatv2_bwd_dst_kernel
#define INST_PN bwd_dst
#else
#define INST_KERNEL gatv2_bwd_src_kernel
#define INST_PN bwd_src
#endif
#define INST_CAT2(a, b, c) gatv2_launch_##a##_##b
#define INST_CAT(a, b) INST_CAT2(a, b, )
#define INST_NAME INST_CAT(INST_PN, INST_TN)

namespace segger {
namespace {

#if SEGGER_INST_PASS == 1
#define INST_LAUNCH(H, LPH, WPR)                                                                                   \
  do {                                                                                                             \
    if (p.direct_gxl)                                                                                              \
      hipLaunchKernelGGL((gatv2_bwd_dst_kernel<INST_T, H, LPH, WPR, true>), dim3((unsigned)p.nblocks_padded),      \
                         dim3(256), 0, stream, p);                                                                 \
    else                                                                                                           \
      hipLaunchKernelGGL((gatv2_bwd_dst_kernel<INST_T, H, LPH, WPR, false>), dim3((unsigned)p.nblocks_padded),     \
                         dim3(256), 0, stream, p);                                                                 \
  } while (0)
#elif SEGGER_INST_PASS == 0 && (defined(EXP_PERSIST) || defined(EXP_EXTRA_STORES))
// bounding builds of the forward (tools/build_variant.sh): a persistent grid and / or a scratch matrix for the extra stores
#ifdef EXP_PERSIST
#define EXP_GRID (p.nblocks_padded < EXP_PERSIST * 256 ? p.nblocks_padded : (int64_t)EXP_PERSIST * 256)
#else
#define EXP_GRID p.nblocks_padded
#endif
#define INST_LAUNCH(H, LPH, WPR)                                                                                   \
  do {                                                                                                             \
    static void* scratch = nullptr; static int64_t scratch_rows = 0;                                               \
    if (scratch_rows < p.n_rows) { (void)hipMalloc(&scratch, (size_t)p.n_rows * 3 * H * LPH * 8 * 4); scratch_rows = p.n_rows; } \
    p.gxl = scratch; p.ld_gxl = 3 * H * LPH * 8;                                                                   \
    hipLaunchKernelGGL((INST_KERNEL<INST_T, H, LPH, WPR>), dim3((unsigned)(EXP_GRID)), dim3(256), 0, stream, p);  \
  } while (0)
#elif SEGGER_INST_PASS == 0 && SEGGER_INST_DTYPE != 0
// 16-bit forward: the LDS-gather kernel where the view carries block tables (flagship geometry, no attention output)
#define INST_LAUNCH(H, LPH, WPR)                                                                                       \
  do {                                                                                                                 \
    if constexpr (H == 2 && LPH == 8 && !WPR) {                                                                        \
      if (p.blk_cnt && !p.alpha && !p.order) {                                                                         \
        hipLaunchKernelGGL((gatv2_fwd_lds_kernel<INST_T, H, LPH>), dim3((unsigned)p.nblocks_padded), dim3(256), 0, stream, p); \
        break;                                                                                                         \
      }                                                                                                                \
    }                                                                                                                  \
    hipLaunchKernelGGL((INST_KERNEL<INST_T, H, LPH, WPR>), dim3((unsigned)p.nblocks_padded), dim3(256), 0, stream, p); \
  } while (0)
#else
#define INST_LAUNCH(H, LPH, WPR) \
  hipLaunchKernelGGL((INST_KERNEL<INST_T, H, LPH, WPR>), dim3((unsigned)p.nblocks_padded), dim3(256), 0, stream, p)
#endif

template <int H, int LPH, bool WPR>
int launch_one(GatParams& p, hipStream_t stream) {
  using G = Geo<H, LPH>;
  const int64_t rows_per_wave = WPR ? 1 : G::NG;
  const int iters = (SEGGER_INST_PASS == 1) ? p.rows_per_wave_iter : 1;
  const int64_t rows_per_block = 4 * rows_per_wave * iters;
  p.nblocks = (p.n_rows + rows_per_block - 1) / rows_per_block;
  p.nblocks_padded = pad_to_xcd(p.nblocks);
  if (p.nblocks == 0) return SEGGER_OK;
  if (p.nblocks_padded > 0x7fffffffLL) {
    set_error("gatv2: %lld rows need more than 2^31 blocks", (long long)p.n_rows);
    return SEGGER_EUNSUPPORTED;
  }
  INST_LAUNCH(H, LPH, WPR);
  SEGGER_LAUNCH_CHECK("gatv2 kernel launch");
  return SEGGER_OK;
}

#if SEGGER_INST_PASS == 0
template <int H, int LPH>
int launch_pair(GatParams& a, GatParams& b, hipStream_t stream) {
  using G = Geo<H, LPH>;
  a.nblocks = (a.n_rows + 4 * G::NG - 1) / (4 * G::NG);           // group-per-row
  b.nblocks = (b.n_rows + 3) / 4;                                   // wave-per-row
  a.nblocks_padded = pad_to_xcd(a.nblocks);
  b.nblocks_padded = pad_to_xcd(b.nblocks);
  const int64_t grid = a.nblocks_padded + b.nblocks_padded;
  if (grid == 0) return SEGGER_OK;
  if (grid > 0x7fffffffLL) { set_error("gatv2: too many blocks"); return SEGGER_EUNSUPPORTED; }
  hipLaunchKernelGGL((gatv2_fwd_pair_kernel<INST_T, H, LPH>), dim3((unsigned)grid), dim3(256), 0, stream, a, b);
  SEGGER_LAUNCH_CHECK("gatv2 pair launch");
  return SEGGER_OK;
}
#endif

}  // namespace

#if SEGGER_INST_PASS == 2
namespace {
template <int H, int LPH>
int launch_src_dst_pair(GatParams& a, GatParams& b, hipStream_t stream) {
  using G = Geo<H, LPH>;
  a.nblocks = (a.n_rows + 4 * G::NG - 1) / (4 * G::NG);           // source pass, group-per-row, one row batch per wave
  const int64_t per_b = 4 * (int64_t)b.rows_per_wave_iter;        // one-pass destination pass, wave-per-row
  b.nblocks = (b.n_rows + per_b - 1) / per_b;
  a.nblocks_padded = pad_to_xcd(a.nblocks);
  b.nblocks_padded = pad_to_xcd(b.nblocks);
  const int64_t grid = a.nblocks_padded + b.nblocks_padded;
  if (grid == 0) return SEGGER_OK;
  if (grid > 0x7fffffffLL) { set_error("gatv2: too many blocks"); return SEGGER_EUNSUPPORTED; }
  hipLaunchKernelGGL((gatv2_bwd_src_dst_pair_kernel<INST_T, H, LPH>), dim3((unsigned)grid), dim3(256), 0, stream, a, b);
  SEGGER_LAUNCH_CHECK("gatv2 backward pair launch");
  return SEGGER_OK;
}
}  // namespace
#define INST_BPAIR_CAT2(a) gatv2_launch_bwd_src_dst_pair_##a
#define INST_BPAIR_CAT(a) INST_BPAIR_CAT2(a)
int INST_BPAIR_CAT(INST_TN)(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream) {
#define X(H, LPH) if (heads == H && channels == LPH * 8) return launch_src_dst_pair<H, LPH>(a, b, stream);
  SEGGER_GEOMETRIES(X)
#undef X
  set_error("gatv2: heads=%d channels=%d has no specialised kernel", heads, channels);
  return SEGGER_EUNSUPPORTED;
}
#endif

#if SEGGER_INST_PASS == 0
#define INST_PAIR_CAT2(a) gatv2_launch_fwd_pair_##a
#define INST_PAIR_CAT(a) INST_PAIR_CAT2(a)
int INST_PAIR_CAT(INST_TN)(GatParams& a, GatParams& b, int heads, int channels, hipStream_t stream) {
#define X(H, LPH) if (heads == H && channels == LPH * 8) return launch_pair<H, LPH>(a, b, stream);
  SEGGER_GEOMETRIES(X)
#undef X
  set_error("gatv2: heads=%d channels=%d has no specialised kernel", heads, channels);
  return SEGGER_EUNSUPPORTED;
}
#endif

int INST_NAME(GatParams& p, int heads, int channels, bool wpr, hipStream_t stream) {
#define X(H, LPH)                                      \
  if (heads == H && channels == LPH * 8)               \
    return wpr ? launch_one<H, LPH, true>(p, stream) : launch_one<H, LPH, false>(p, stream);
  SEGGER_GEOMETRIES(X)
#undef X
  set_error("gatv2: heads=%d channels=%d has no specialised kernel (supported: channels in {32,64}, heads in {1,2,3,4})",
            heads, channels);
  return SEGGER_EUNSUPPORTED;
}

}  // namespace segger
