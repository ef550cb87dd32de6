/*
 * segger_amd.h -- C ABI of libsegger_amd.so: hand-written gfx950 (MI355X) HIP
 * kernels for segger's GNN hot path (heterogeneous GATv2 message passing over
 * transcript<->boundary graphs + the transcript->cell edge-scoring heads).
 *
 * The reference (dpeerlab/segger) has no FFI for this path: the arithmetic is
 * reached through Python calls into torch_geometric / torch_scatter / torch
 * CUDA kernels.  Each entry point below names the reference call site (file:line
 * under /root/reference) whose third-party kernel(s) it replaces.
 *
 * Conventions (every entry point):
 *   - returns SEGGER_OK (0) or a negative SEGGER_E* code; segger_last_error()
 *     returns a thread-local message for the last failure on this thread;
 *   - never throws, never allocates or frees device memory, never synchronises:
 *     work is only enqueued on `stream` (a hipStream_t; NULL = default stream);
 *   - all pointers are DEVICE pointers borrowed for the duration of the enqueued
 *     work; outputs and workspaces are caller-allocated
 *     (sizes from the matching *_workspace_bytes());
 *   - sizes / offsets / leading dimensions are int64_t (cf. the >2^31 element
 *     hazard in reference src/segger/_patches.py:1-9); node ids inside CSR
 *     arrays are int32_t, so one batch holds < 2^31 nodes and < 2^31 edges
 *     per edge type;
 *   - "ld_*" = row stride in ELEMENTS of the tensor's dtype; rows must be
 *     16-byte aligned (ld * sizeof(elem) % 16 == 0 and base % 16 == 0);
 *   - feature tensors are row-major [n, heads*channels] in `dtype`
 *     (SEGGER_F32 / SEGGER_BF16 / SEGGER_F16); all accumulation is fp32;
 *     parameters (att, bias) and their gradients are fp32.
 */
#ifndef SEGGER_AMD_H
#define SEGGER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only what is declared between this push and the pop at the end of the
 * header -- the C entry points -- reaches its dynamic symbol table (a definition takes the visibility of its first
 * declaration); the C++ launchers behind them (namespace segger) stay internal. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SEGGER_ABI_VERSION 30

enum segger_status {
  SEGGER_OK = 0,
  SEGGER_EINVAL = -1,       /* bad argument (null pointer, negative size, misaligned ld) */
  SEGGER_EUNSUPPORTED = -2, /* valid but not implemented (heads/channels combination, dtype) */
  SEGGER_EHIP = -3,         /* a HIP runtime call failed; message has hipGetErrorString */
  SEGGER_EWORKSPACE = -4    /* workspace too small */
};

enum segger_dtype { SEGGER_F32 = 0, SEGGER_BF16 = 1, SEGGER_F16 = 2 };

typedef void* segger_stream_t; /* hipStream_t */

int segger_abi_version(void);
const char* segger_last_error(void);

/* ------------------------------------------------------------------------
 * Compressed sparse rows of one edge type.
 *   rows = the node the kernel iterates over, col = the node it gathers.
 *   eid[slot] = position of that edge in the caller's COO edge_index, used
 *   (a) for outputs indexed by original edge id, (b) as the dropout counter so
 *   forward, dst-side backward and src-side backward draw the same mask.
 * ---------------------------------------------------------------------- */
typedef struct segger_csr {
  const int64_t* indptr; /* [n_rows + 1] */
  const int32_t* col;    /* [n_edges] */
  const int32_t* eid;    /* [n_edges] */
  int64_t n_rows;
  int64_t n_cols;
  int64_t n_edges;
  const int32_t* row_order; /* [n_rows] or NULL: the order in which kernels visit the rows (a permutation of
                               0..n_rows-1 from segger_csr_row_order); results never depend on it */
  /* optional block tables (segger_csr_block_tables; all three or none): per workgroup of SEGGER_BLOCK_ROWS consecutive
   * visiting positions the DISTINCT column ids its rows gather -- staged in LDS once, then gathered from there */
  const int32_t* blk_cnt;   /* [ceil(n_rows / SEGGER_BLOCK_ROWS)]: number of distinct ids, <= SEGGER_BLOCK_CAP */
  const int32_t* blk_src;   /* [n_blocks][SEGGER_BLOCK_CAP]: the ids (slot order arbitrary) */
  const uint8_t* col_local; /* [n_edges]: per CSR slot, the LDS slot of col[slot] in its block's table */
} segger_csr;
#define SEGGER_BLOCK_ROWS 16
#define SEGGER_BLOCK_CAP 128
/*
 * segger_csr_block_tables: the block tables of a CSR view in a given visiting order (row_order NULL: natural).  A kNN graph
 * over spatially sorted nodes re-uses its gathered rows heavily inside a workgroup (C2 tile: the 16 rows of a workgroup
 * gather 240 rows of which 58 are distinct), so the aggregation kernels can load each distinct row ONCE per workgroup,
 * coalesced, into LDS and gather from there instead of issuing 240 row gathers through the texture addresser.
 * *overflow (device int32, zeroed by the caller) is set when some block has more than SEGGER_BLOCK_CAP distinct ids or more
 * than 1024 edges: the tables are then unusable and the caller keeps the plain kernels.
 */
int segger_csr_block_tables(const int64_t* indptr, const int32_t* col, const int32_t* row_order, int64_t n_rows, int64_t n_edges,
                            int32_t* blk_cnt, int32_t* blk_src, uint8_t* col_local, int32_t* overflow, segger_stream_t stream);

/*
 * segger_csr_from_coo: stable sort of COO edges by `row`, producing indptr /
 * col / eid.  Replaces what PyG does implicitly per layer with scatter
 * kernels keyed on edge_index[1] (GATv2Conv.propagate; constructed at
 * src/segger/models/ist_encoder.py:111-124) and torch_scatter.scatter_max's
 * keying on edge_index[0] (src/segger/models/lightning_model.py:280-284):
 * built ONCE per batch and shared by all layers, forward and backward.
 *   row, colv : [n_edges] int64 (edge_index rows as PyG stores them)
 *   n_invalid : optional device int32; receives the number of edges whose row
 *               or col id is out of range (such ids are clamped to 0 so later
 *               kernels cannot fault; the host wrapper raises on n_invalid>0).
 */
size_t segger_csr_from_coo_workspace_bytes(int64_t n_edges, int64_t n_rows);
int segger_csr_from_coo(const int64_t* row, const int64_t* colv, int64_t n_edges,
                        int64_t n_rows, int64_t n_cols,
                        int64_t* indptr, int32_t* col, int32_t* eid,
                        int32_t* n_invalid,
                        void* workspace, size_t workspace_bytes,
                        segger_stream_t stream);

/*
 * segger_csr_row_order: a visiting order for the rows of a CSR that balances the lanes of a wavefront.
 * The aggregation kernels give each 16-lane group of a wave its own row; a wave is busy until its
 * longest row is done, so rows of unequal degree (kNN in-degrees: mean k, spread ~ sqrt(k)) idle lanes.
 * Inside every window of `window` consecutive rows (a power of two <= 64, so neighbouring rows still
 * share an L2) the rows are ordered by descending degree: the 4 rows of a wave then have near-equal
 * length.  order_out[n_rows] is a permutation of 0..n_rows-1.
 */
int segger_csr_row_order(const int64_t* indptr, int64_t n_rows, int32_t window,
                         int32_t* order_out, segger_stream_t stream);

/* ------------------------------------------------------------------------
 * GATv2 attention aggregation (the roofline kernel).
 *
 * Replaces torch_geometric.nn.GATv2Conv.edge_update + utils.softmax + message
 * + aggregate('add') + bias (PyG 2.7.0), i.e. everything in the conv AFTER
 * lin_l / lin_r, for one edge type of segger's HeteroConv
 * (src/segger/models/ist_encoder.py:109-134,183-189), with the following GELU
 * (ist_encoder.py:325) optionally fused:
 *
 *   e[i->j,h]  = sum_c att[h,c] * leaky_relu(x_l[i,h,c] + x_r[j,h,c], slope)
 *   a[i->j,h]  = softmax over the in-edges of j ;  a *= keep/(1-p) (dropout)
 *   pre[j,h,c] = sum_i a[i->j,h] * x_l[i,h,c] + bias[h,c]
 *   out        = apply_gelu ? gelu_erf(pre) : pre
 *                (exact-erf GELU, torch approximate='none'; evaluated through a polynomial erfc with
 *                |error| <= 4e-7 in fp32, see csrc/common.h::normal_cdf)
 *
 * A destination without in-edges gets pre = bias.
 * Dropout (ist_encoder.py:116,123; training only), with (lo, hi) = halves of splitmix64(seed + *seed_dev):
 *   x = (eid*H + h) ^ lo;  x *= 0x85ebca6b; x ^= x>>13; x *= 0xc2b2ae35; x ^= x>>16; x ^= hi;
 *   keep(e,h) = (x >> 8) >= floor(p * 2^24)      (uint32 arithmetic).
 * ---------------------------------------------------------------------- */
typedef struct segger_gatv2_fwd_args {
  segger_csr by_dst;      /* rows = destination nodes, col = source ids */
  const void* x_l;        /* [n_src, H*C] lin_l(x_src) */
  int64_t ld_xl;
  const void* x_r;        /* [n_dst, H*C] lin_r(x_dst) */
  int64_t ld_xr;
  const float* att;       /* [H*C] */
  const float* bias;      /* [H*C] or NULL */
  int32_t heads;
  int32_t channels;
  int32_t dtype;          /* segger_dtype of x_l, x_r, out, pre */
  int32_t apply_gelu;
  float negative_slope;   /* 0.2 in GATv2Conv */
  float dropout_p;        /* 0 = eval */
  uint64_t seed;
  const uint64_t* seed_dev; /* optional DEVICE word: the effective seed is seed + *seed_dev, read when the kernel
                               runs (a captured hipGraph draws a fresh mask every replay); NULL = seed alone */
  void* out;              /* [n_dst, H*C] */
  int64_t ld_out;
  void* pre;              /* [n_dst, H*C] pre-activation for backward; NULL = skip
                             (may alias out when !apply_gelu) */
  int64_t ld_pre;
  float* lse;             /* [n_dst, H] log2(sum exp2(e*log2(e))) per (dst, head), kernel units;
                             NULL = skip */
  float* alpha;           /* [n_edges, H] attention (after dropout) by ORIGINAL edge id; NULL = skip
                             (SkipGAT.attention_weights, ist_encoder.py:146-158,192-211) */
  const uint8_t* keep_bits; /* optional, [n_edges] in by_dst SLOT order: bit h = keep(e, h) of the mask defined
                               above, precomputed by segger_dropout_bits for this layer's seed.  The kernels then test
                               a bit instead of hashing per (edge, head); results are identical.  Ignored when alpha
                               is requested or the geometry runs on the generic kernels. */
} segger_gatv2_fwd_args;

int segger_gatv2_fwd(const segger_gatv2_fwd_args* args, segger_stream_t stream);
/* Two edge types of one HeteroConv layer (ist_encoder.py:109-134: tx-neighbors-tx and tx-belongs-bd) in ONE launch: the
 * few thousand short blocks of the high-degree edge type `b` (wave per destination row) run beside the blocks of the
 * low-degree edge type `a` (lane group per row) instead of as a latency-bound launch of their own.  Same arguments as
 * two segger_gatv2_fwd calls; falls back to two launches when the pair is not (group-per-row, wave-per-row) of one
 * specialised geometry and storage type, or when attention weights are requested. */
int segger_gatv2_fwd_pair(const segger_gatv2_fwd_args* a, const segger_gatv2_fwd_args* b, segger_stream_t stream);

/*
 * Backward of the above (replaces autograd through the PyG ops).  Two atomic-free
 * passes: a destination-side pass (CSR by dst) producing grad_pre, grad_xr,
 * grad_att, grad_bias, and a source-side pass (CSR by src) producing grad_xl.
 * Attention coefficients are recomputed from x_l, x_r and `lse`; no [E,H,C]
 * tensor is ever materialised.  Outputs are WRITTEN (not accumulated).
 */
typedef struct segger_gatv2_bwd_args {
  segger_csr by_dst;      /* as in forward */
  segger_csr by_src;      /* rows = source nodes, col = destination ids (same edges) */
  const void* x_l;
  int64_t ld_xl;
  const void* x_r;
  int64_t ld_xr;
  const float* att;
  const float* bias;      /* may be NULL */
  int32_t heads;
  int32_t channels;
  int32_t dtype;
  int32_t apply_gelu;
  float negative_slope;
  float dropout_p;
  uint64_t seed;
  const uint64_t* seed_dev; /* as in forward; must hold the value the forward saw */
  const void* grad_out;   /* [n_dst, H*C] dL/d out */
  int64_t ld_go;
  const void* pre;        /* [n_dst, H*C] from forward */
  int64_t ld_pre;
  const float* lse;       /* [n_dst, H] from forward */
  void* grad_pre;         /* [n_dst, H*C] scratch+output: dL/d pre (dtype) */
  int64_t ld_gp;
  float* dsum;            /* [n_dst, H, 2] scratch: (lse, D) pairs, D = sum_c grad_pre * (pre - bias) */
  void* grad_xl;          /* [n_src, H*C] */
  int64_t ld_gxl;
  void* grad_xr;          /* [n_dst, H*C] */
  int64_t ld_gxr;
  float* grad_att;        /* [H*C] */
  float* grad_bias;       /* [H*C] or NULL */
  void* workspace;        /* segger_gatv2_bwd_workspace_bytes() */
  size_t workspace_bytes;
  const uint8_t* keep_bits_dst; /* optional dropout bit planes (see forward) in by_dst / by_src slot order */
  const uint8_t* keep_bits_src;
  int32_t src_unique;     /* non-zero: the caller asserts that no source node has more than one out-edge (segger's
                             tx-belongs-bd: a transcript lies in at most one boundary; check with
                             segger_coo_unique).  grad_xl[i] then has a single term, the destination-side pass
                             stores it itself and by_src is ignored (may be all zero): no by-source sort, no
                             source-side pass.  Needs a specialised geometry (segger_gatv2_has_specialised). */
  void* zero_rows_out;    /* optional, two-pass form only: a second [n_src, H*C] matrix (row stride ld_zero) that the
                             source pass zero-fills row by row while it writes grad_xl -- the grad_xl of ANOTHER edge
                             type over the same source nodes whose one-pass backward runs next (segger: tx-neighbors-tx
                             fills the tx-belongs-bd window of the stacked projection gradient) */
  int64_t ld_zero;
  int32_t grad_xl_zeroed; /* one-pass form (src_unique): grad_xl already holds zeros (see zero_rows_out): skip the fill */
  int32_t passes;         /* segger_gatv2_bwd only -- 0: the whole backward (default); 1: the destination-side pass alone
                             (grad_pre, dsum, grad_xr, grad_att, grad_bias; grad_xl too in the one-pass form); 2: the
                             source-side pass alone (grad_xl from the grad_pre / dsum an earlier passes = 1 call left in
                             the same buffers).  1 then 2 == 0; exists so that a profiler / bench.py can time the two
                             kernels of the backward separately (roofline.dominant).  3: the destination KERNEL alone --
                             as 1 without the two small launches that sum its per-workgroup grad_att / grad_bias partials
                             (those two outputs are then not written): timing only. */
} segger_gatv2_bwd_args;

size_t segger_gatv2_bwd_workspace_bytes(int64_t n_dst, int32_t heads, int32_t channels);
int segger_gatv2_bwd(const segger_gatv2_bwd_args* args, segger_stream_t stream);
/* The backward of both edge types of one HeteroConv layer (ist_encoder.py:109-134 under autograd): `a` a two-pass edge
 * type (tx-neighbors-tx), `b` a one-pass one (src_unique: tx-belongs-bd) whose grad_xl is the matrix `a` zero-fills
 * (a->zero_rows_out == b->grad_xl with b->grad_xl_zeroed set, or neither set).  Same results as segger_gatv2_bwd(a)
 * followed by segger_gatv2_bwd(b).  For batches of at most SEGGER_BWD_PAIR_MAX_ROWS source nodes (environment
 * override of the same name, read once) of one specialised geometry and storage type, with a's destinations and b's
 * sources indexing the same nodes, the zero fill moves into a's DESTINATION pass and a's source pass shares ONE launch
 * with b's destination pass (b's few hundred latency-bound blocks run beside a's); otherwise the two calls run as they
 * are. */
int segger_gatv2_bwd_pair(const segger_gatv2_bwd_args* a, const segger_gatv2_bwd_args* b, segger_stream_t stream);
/*
 * segger_dropout_bits: the attention-dropout mask of n_seeds layers as bit planes over the slots of one CSR view:
 *   bits[l * plane_stride + slot] = sum_h keep(eid[slot], h; seeds[l] + *seed_dev) << h        (heads <= 8)
 * with keep() exactly the counter-based mask of segger_gatv2_fwd.  One launch per view per training step replaces
 * 3 passes x n_seeds layers of per-(edge, head) hashing inside the aggregation kernels.  seeds is a HOST array.
 * plane_stride: bytes between the planes of consecutive layers, a multiple of 4 and >= n_edges (a thread writes four
 * slots of a plane as one word); bits 4-byte aligned.
 */
int segger_dropout_bits(const int32_t* eid, int64_t n_edges, int32_t heads, float dropout_p, const uint64_t* seeds,
                        int32_t n_seeds, const uint64_t* seed_dev, uint8_t* bits, int64_t plane_stride,
                        segger_stream_t stream);
/* segger_dropout_bits_many: up to 4 such views (the by-destination / by-source views of a step's edge types) in ONE
 * launch; per view the arguments of segger_dropout_bits.
 * segger_step_advance: *step += inc and (optionally) *copy = the new value, one tiny launch -- the device-side step
 * counter behind seed_dev (ISTEncoder's dropout stream), advanced and snapshotted for the backward. */
typedef struct segger_bits_job {
  const int32_t* eid;
  int64_t n_edges;
  const uint64_t* seeds;    /* HOST array [n_seeds] */
  int32_t n_seeds;
  int32_t reserved_;
  uint8_t* bits;
  int64_t plane_stride;
} segger_bits_job;
int segger_dropout_bits_many(const segger_bits_job* jobs, int32_t n_jobs, int32_t heads, float dropout_p,
                             const uint64_t* seed_dev, segger_stream_t stream);
int segger_step_advance(int64_t* step, int64_t inc, int64_t* copy, segger_stream_t stream);

/* segger_step_draws: ALL random draws of one training step in one launch -- the bit planes of up to four edge views
 * (segger_dropout_bits_many), up to two cluster-aware triplet samplers (segger_triplet_sample: loss_tx and loss_bd,
 * models/triplet_loss.py:83-125, with the kernel's own counter-based uniforms) and the negative boundaries of the
 * segmentation loss (segger_sample_negatives, lightning_model.py:167-176).  Same streams, same results as the separate
 * calls with the same seeds; every job reads the device word seed_dev. */
typedef struct segger_sample_job {
  const int64_t* lab; int64_t n; int32_t n_clusters; int32_t reserved_;
  const float* cdf_pos; const float* cdf_neg; const int64_t* counts; const int64_t* offsets; const int64_t* members;
  uint64_t seed; const float* dists; int64_t* pos; int64_t* neg; float* d_pos; float* d_neg;
} segger_sample_job;
typedef struct segger_step_draws_args {
  const segger_bits_job* bits; int32_t n_bits; int32_t heads; float dropout_p; int32_t n_samplers;
  segger_sample_job samplers[2];
  const int64_t* neg_pos; int64_t neg_n; int64_t neg_n_b; const int64_t* neg_n_b_dev; uint64_t neg_seed; int64_t* neg_out;
  const uint64_t* seed_dev;
  float* const* advance; int32_t n_advance; int32_t reserved_;   /* HOST array of up to 64 device floats: *advance[i] += 1 (Adam's
                                                                    step counters, see segger_adam_step_ex) */
} segger_step_draws_args;
int segger_step_draws(const segger_step_draws_args* args, segger_stream_t stream);

/* 1 when (heads, channels) runs on the specialised kernels (channels in {32,64}, heads in 1..4), 0 = generic kernels */
int segger_gatv2_has_specialised(int32_t heads, int32_t channels);

/*
 * segger_coo_unique: *unique_out = 1 when no value occurs twice in ids[n] (values in [0, n_ids)), else 0.
 * marks: caller-provided int32[n_ids] scratch (zeroed by the call).  Enqueue-only; read unique_out after the stream
 * has passed.  Used once per batch on edge_index[0] of tx-belongs-bd (src/segger/data/utils/heterodata.py:147: each
 * transcript is assigned to at most one boundary) to license src_unique above.
 */
int segger_coo_unique(const int64_t* ids, int64_t n, int64_t n_ids, int32_t* marks, int32_t* unique_out,
                      segger_stream_t stream);

/*
 * segger_stage: n_segs independent "copy a prefix, fill the rest" jobs in one launch -- how a batch is written into the
 * static, padded buffers a captured training step (hipGraph) replays on (no counterpart in the reference, which
 * re-launches every op of every step; lightning_model.py:151-231 is what the captured step computes).
 * Element i of segment s:   i < n_copy: src[i] (read as src_bytes-wide integer, stored dst_bytes-wide; equal sizes
 * copy bit patterns, so floats pass unchanged);  otherwise with k = i - n_copy:
 *   SEGGER_FILL_CONST  a                       (bit pattern of the value, e.g. 0 for 0.0f)
 *   SEGGER_FILL_TILE   src[k % a]              (replicate the first a elements: "copies of row 0")
 *   SEGGER_FILL_DIV    a + k / b               (b = 1: an iota)
 *   SEGGER_FILL_MOD    a + k % b
 *   SEGGER_FILL_RAMP   a + min((k + 1) * b, c) (row pointers of rows holding b padding edges each, c in total)
 * segs is a HOST array (copied into the kernel arguments).
 */
enum { SEGGER_FILL_CONST = 0, SEGGER_FILL_TILE = 1, SEGGER_FILL_DIV = 2, SEGGER_FILL_MOD = 3, SEGGER_FILL_RAMP = 4 };
typedef struct {
  void* dst;
  const void* src;
  int64_t n_copy, n_total;   /* elements */
  int64_t a, b, c;           /* fill parameters */
  int32_t dst_bytes, src_bytes;
  int32_t fill;
  int32_t copy_add;          /* added to every copied and TILE-replicated element (INTEGER segments only), e.g. to
                                rebase ids while they are copied; 0 for float segments */
} segger_stage_seg;
int segger_stage(const segger_stage_seg* segs, int32_t n_segs, segger_stream_t stream);

/*
 * segger_adam_step: torch.optim.Adam.step() (lightning_model.py:300-303: Adam(lr), default betas / eps, no weight decay,
 * no amsgrad) for n_tensors fp32 parameter tensors at once, on the optimizer's OWN state tensors (exp_avg, exp_avg_sq and
 * the per-tensor fp32 step counter of a capturable torch Adam), so checkpoints and an eager optimizer.step() stay
 * interchangeable:   step += 1;  exp_avg = lerp(exp_avg, grad, 1 - beta1);  exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) grad^2;
 * param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps).
 * Two launches per 64 tensors (counters, then all elements) instead of torch's three multi-tensor launches per step.
 * lr / betas / eps are doubles as torch holds them (1 - beta and beta^step are evaluated in double, the rest in fp32).
 * tensors is a HOST array (copied into the kernel arguments); zero-sized tensors only advance their counter.
 */
typedef struct segger_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* step;               /* device scalar, fp32 (torch's capturable step counter) */
  int64_t numel;
} segger_adam_tensor;
int segger_adam_step(const segger_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps,
                     segger_stream_t stream);
/* segger_adam_step_ex: the same with (a) flags & SEGGER_ADAM_STEPS_ADVANCED: the step counters have been advanced already
 * (segger_step_draws does it at the head of a captured training step: nothing in between reads them) -- one launch instead
 * of two; (b) counter != NULL: *counter += counter_inc by the update launch's first workgroup -- the device-side dropout /
 * sampling counter of the step that ends here (nothing reads it after the backward). */
#define SEGGER_ADAM_STEPS_ADVANCED 1
int segger_adam_step_ex(const segger_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps,
                        int32_t flags, int64_t* counter, int64_t counter_inc, segger_stream_t stream);
/* segger_adam_step_dev: segger_adam_step_ex with {lr, beta1, beta2, eps} read from a DEVICE array of four doubles when the
 * kernel runs (not validated: the caller writes what its optimizer's param_group holds).  A captured training step keeps the
 * array in its static buffers and refreshes it with the batch (segger_stage), so a learning-rate scheduler that changes
 * `param_groups[0]["lr"]` every step (lightning_model.py:300-303 returns a bare Adam; schedulers are the user's) is followed
 * by the replayed graph without a new capture. */
int segger_adam_step_dev(const segger_adam_tensor* tensors, int32_t n_tensors, const double* hyper, int32_t flags,
                         int64_t* counter, int64_t counter_inc, segger_stream_t stream);

/* segger_transpose_many: dst [cols, rows] = src [rows, cols]^T for n_segs contiguous 16-bit matrices in one launch:
 * the W^T copies the data-gradient GEMMs (dX = dY W on segger_linear_fwd) need after every optimizer step. */
typedef struct {
  void* dst;
  const void* src;
  int32_t rows, cols;
} segger_transpose_seg;
int segger_transpose_many(const segger_transpose_seg* segs, int32_t n_segs, segger_stream_t stream);
/* segger_pack_refresh: the compute-dtype copies of fp32 master weights after an optimizer step, all in ONE launch.
 * Per segment: src fp32 [rows, cols] row-major -> dst (bf16 / f16 per `dtype`, [rows, cols] contiguous: the matrix's row
 * window of a row-stacked buffer) and, when dst_t != NULL, its transpose into dst_t[c * ld_t + r] (dst_t already points
 * at the matrix's COLUMN window of the stacked transposed buffer [cols, ld_t]).  dst_f32 != 0: a bias -- `rows` floats
 * copied fp32 -> fp32 (cols, dst_t ignored).  Replaces torch._foreach_copy_ x 2 + segger_transpose_many. */
typedef struct {
  const float* src;
  void* dst;
  void* dst_t;
  int64_t ld_t;
  int32_t rows, cols;
  int32_t dst_f32;
  int32_t reserved_;
} segger_pack_seg;
int segger_pack_refresh(const segger_pack_seg* segs, int32_t n_segs, int32_t dtype, segger_stream_t stream);

/* ------------------------------------------------------------------------
 * Prediction head: cosine similarity on tx->bd candidate edges + per-transcript
 * arg-max + assignment.  Replaces torch.cosine_similarity on two gathered
 * [Ep, C] tensors, torch_scatter.scatter_max and the masked fancy-indexing of
 * src/segger/models/lightning_model.py:275-293.
 *   ties -> lowest original edge id (torch_scatter CPU semantics);
 *   transcript without candidate edges -> max_sim 0, max_eid n_edges, seg -1.
 * ---------------------------------------------------------------------- */
typedef struct segger_edge_argmax_args {
  segger_csr by_src;      /* rows = transcripts, col = boundary ids */
  const void* z_src;      /* [n_rows, C] */
  int64_t ld_zs;
  const void* z_dst;      /* [n_cols, C] */
  int64_t ld_zd;
  int32_t channels;
  int32_t dtype;
  float eps;              /* 1e-8 (torch.cosine_similarity) */
  int32_t use_min_similarity;
  float min_similarity;
  const int64_t* dst_index; /* [n_cols] boundary 'index' attribute; NULL = identity */
  float* max_sim;         /* [n_rows] */
  int64_t* max_eid;       /* [n_rows] */
  int64_t* seg_idx;       /* [n_rows] dst_index[col of arg-max] or -1 */
  float* sim;             /* [n_edges] per-edge cosine by ORIGINAL edge id; NULL = skip */
} segger_edge_argmax_args;

int segger_edge_cos_argmax(const segger_edge_argmax_args* args, segger_stream_t stream);

/* ------------------------------------------------------------------------
 * Training head: triplet margin loss over tx-belongs-bd edges with sampled
 * negative boundaries.  Replaces 3 gathers + torch.nn.TripletMarginLoss
 * (margin, p=2, eps=1e-6, mean) at src/segger/models/lightning_model.py:182-187
 * and its autograd.
 *   loss = mean_e max(||a-p+eps|| - ||a-n+eps|| + margin, 0)
 * fwd writes partial sums; `loss` receives the mean (loss == NULL: partial sums only, see
 * segger_loss_combine_partials_fwd).  bwd accumulates
 * (atomics: fp32, or packed 16-bit pairs, see grad_*_packed) into grad_a / grad_b, which the caller zero-fills.
 * ---------------------------------------------------------------------- */
typedef struct segger_triplet_args {
  const int64_t* src;     /* [n_edges] anchor rows of z_a */
  const int64_t* pos;     /* [n_edges] positive rows of z_b */
  const int64_t* neg;     /* [n_edges] negative rows of z_b */
  int64_t n_edges;
  const void* z_a;        /* [n_a, C] */
  int64_t ld_za;
  int64_t n_a;
  const void* z_b;        /* [n_b, C] */
  int64_t ld_zb;
  int64_t n_b;
  int32_t channels;
  int32_t dtype;
  float margin;
  float eps;
  float* loss;            /* [1] */
  float grad_scale;       /* bwd: dL/dloss (the 1/n_edges is applied inside) */
  const float* grad_scale_dev; /* bwd: optional DEVICE scalar multiplied into grad_scale (avoids a host sync) */
  void* grad_a;           /* [n_a, C] bwd only: fp32, or `dtype` when grad_a_packed */
  void* grad_b;           /* [n_b, C] bwd only: fp32, or `dtype` when grad_b_packed */
  int32_t grad_a_packed;  /* 1: accumulate with packed 2-channel atomics in the embeddings' 16-bit dtype (half the
                             atomics, no fp32 staging + cast); meant for matrices whose rows collect a handful of
                             terms (transcripts); needs a 16-bit dtype and even C */
  int32_t grad_b_packed;
  float* contrib;         /* bwd, optional: [n_edges, 2, C] fp32.  When given, the z_b side uses NO atomics: triplet e
                             leaves (-dL/d z_b[pos_e], +dL/d z_b[neg_e]) in rows 2e, 2e+1 (zeros when inactive) and the
                             caller sums the rows grouped by boundary with segger_segment_rowsum; grad_b is ignored */
  void* workspace;
  size_t workspace_bytes;
  const int64_t* pos_indptr; /* bwd, optional: [n_b + 1] and ... */
  const int32_t* pos_eid;    /* ... [n_edges]: the triplets grouped by their positive row (for tx-belongs-bd edges this is
                                the by-destination CSR the encoder already built).  When given (separate z_b, fp32
                                grad_b), the positive side of grad_b is a segmented sum over these groups -- written,
                                not accumulated, so grad_b needs no zero-fill -- and only the negatives, which are
                                sampled uniformly and therefore uncontended, use fp32 atomics.  (Positives are the hot
                                rows: a boundary's ~40 edges sit next to each other and hammer one row.) */
  int32_t anchor_unique;     /* bwd, with pos_indptr: 1 = no row of z_a is the anchor of two triplets (tx-belongs-bd: a
                                transcript lies in at most one boundary, heterodata.py:147).  The whole backward is then
                                ONE walk over the groups: the anchor's row of grad_a is STORED (non-anchor rows keep the
                                caller's zeros; later kernels may add to it), the group's positive row is summed in
                                registers and added once, negatives add by fp32 atomics -- grad_b is ACCUMULATED INTO
                                (caller zero-fills it), every embedding row is read once.  C in {32, 64, 96, 128}. */
  int32_t loss_kind;         /* SEGGER_LOSS_TRIPLET (0) or SEGGER_LOSS_BCE: the BCE variant of the segmentation loss
                                (lightning_model.py:190-207), BCEWithLogits over the dot-product logits <a, pos> (label 1)
                                and <a, neg> (label 0), mean over the 2 n_edges logits; margin / eps / contrib unused;
                                grad_b is always accumulated into (caller zero-fills); with pos_indptr + anchor_unique
                                the backward is the same one walk over the groups as for the triplet loss */
  void* grad_a_rows;         /* bwd, optional (triplet loss): [n_a, C] in the format of grad_a.  For triplets whose anchor is
                                their own index (src[e] == e or skipped, n_edges == n_a: loss_tx, where every row anchors one
                                triplet) the anchor's term of triplet e is STORED into row e (zeros when the triplet is
                                inactive or skipped) instead of added atomically; grad_a / grad_b then receive only the
                                positive / negative terms.  The consumer adds the two matrices (segger_l2norm_bwd2). */
} segger_triplet_args;
#define SEGGER_LOSS_TRIPLET 0
#define SEGGER_LOSS_BCE 1

/*
 * segger_loss_combine_fwd / _bwd: the loss combination of LitISTEncoder.get_losses (lightning_model.py:136-149,
 * 210-211: scheduled weights, weighted sum) on DEVICE scalars, one launch each way instead of a chain of 0-dim torch ops:
 *   out[i] = raw[i] * a[i]  (the logged losses; a = per-loss rescaling, e.g. padded -> masked means),
 *   out[n] = sum_i out[i] * b[i]  (b = scheduled weights);    grad_raw[i] = (grad_out[n] * b[i] + grad_out[i]) * a[i].
 * raw / a / b: float[n] on the device, n <= 16; grad_raw feeds grad_scale_dev of the loss kernels' backward.
 */
int segger_loss_combine_fwd(const float* raw, const float* a, const float* b, int32_t n, float* out, segger_stream_t stream);
/* The same with raw[i] = scale[i] * sum(partial[i][0 .. n_partial[i])): segger_triplet_fwd / segger_metric_fwd called with
 * loss == NULL leave their per-block partial sums (segger_triplet_partial_count(n) floats) at the start of their workspace;
 * this launch finishes all of them (in a fixed order) and combines -- one launch instead of one per loss + one.
 * partial / n_partial / scale are HOST arrays of n entries (device pointers inside partial). */
int64_t segger_triplet_partial_count(int64_t n_edges);
int segger_loss_combine_partials_fwd(const float* const* partial, const int64_t* n_partial, const float* scale, const float* a,
                                     const float* b, int32_t n, float* out, segger_stream_t stream);
int segger_loss_combine_bwd(const float* grad_out, const float* a, const float* b, int32_t n, float* grad_raw,
                            segger_stream_t stream);

/*
 * segger_triplet_sample: FastTripletSelector.sample_triplets (src/segger/models/triplet_loss.py:83-125) for all
 * nodes in one launch: per node i with cluster r = lab[i], draw the positive cluster from row r of cdf_pos and the
 * negative cluster from row r of cdf_neg (first column whose cumulative probability >= u, as torch.searchsorted),
 * then the member  members[offsets[c] + floor(u' * counts[c])]  of the drawn cluster c.
 *   lab      [n] int64 cluster id per node; ids >= n_clusters (the "masked out" cluster) get pos = neg = -1
 *   cdf_pos / cdf_neg [n_clusters, n_clusters] fp32 row-wise cumulative distributions
 *   counts / offsets  [n_clusters (+1)] int64, members [n] int64 (nodes grouped by cluster)
 *   uniforms [4, n] fp32 (cluster+, member+, cluster-, member-) or NULL: then U[0,1) comes from a counter-based
 *            generator keyed by (seed + *seed_dev, node, draw) -- same construction as the dropout stream
 *   dists    [n_clusters, n_clusters] fp32 or NULL; when given d_pos / d_neg [n] receive dists[r, cluster of pick]
 */
int segger_triplet_sample(const int64_t* lab, int64_t n, int32_t n_clusters, const float* cdf_pos, const float* cdf_neg,
                          const int64_t* counts, const int64_t* offsets, const int64_t* members,
                          const float* uniforms, uint64_t seed, const uint64_t* seed_dev,
                          const float* dists, int64_t* pos, int64_t* neg, float* d_pos, float* d_neg,
                          segger_stream_t stream);

/*
 * segger_sample_negatives: the negative destinations of the segmentation loss (lightning_model.py:178-180:
 *   dst_neg = (dst_pos + randint(1, n_b)) % n_b ) in one launch:  neg[e] = (pos[e] + 1 + floor(u * (n_b - 1))) % n_b
 * with u from the counter-based generator keyed by (seed + *seed_dev, e); pos[e] < 0 (a padded triplet) gives -1.
 * n_b_dev (optional) overrides n_b with a device value (a captured step padded to a static size); n_b <= 1 gives 0.
 */
int segger_sample_negatives(const int64_t* pos, int64_t n, int64_t n_b, const int64_t* n_b_dev, uint64_t seed,
                            const uint64_t* seed_dev, int64_t* neg, segger_stream_t stream);

/*
 * segger_metric_fwd / _bwd: MetricLoss.forward (src/segger/models/triplet_loss.py:163-204) on sampled triplets,
 *   loss = sum_i w_i * [ (cos(z_i, z_pos_i) - (1 - d_pos_i))^2 + (cos(z_i, z_neg_i) - (1 - d_neg_i))^2 ]
 * with cos(x, y) = sum (x / max(|x|, eps)) * (y / max(|y|, eps)) (torch.cosine_similarity, eps = 1e-8) and w_i the
 * weight of node i (1/n for the reference's plain mean; mask_i / #masked for the masked form).  pos_i / neg_i < 0
 * (or w_i == 0) skip the node.  ~40 torch launches forward and as many backward become one kernel each.
 *   fwd: partial sums in workspace (segger_triplet_workspace_bytes(n)), mean in loss[0].
 *   bwd: grad_z [n, C] fp32, ZERO-FILLED by the caller, receives d loss / d z scaled by grad_scale_dev[0]
 *        (own row: plain add by the owning lanes; positive / negative rows: fp32 atomics).
 */
int segger_metric_fwd(const void* z, int64_t ld_z, int64_t n, int32_t channels, int32_t dtype, const int64_t* pos,
                      const int64_t* neg, const float* d_pos, const float* d_neg, const float* w, float eps,
                      float* loss, void* workspace, size_t workspace_bytes, segger_stream_t stream);
int segger_metric_bwd(const void* z, int64_t ld_z, int64_t n, int32_t channels, int32_t dtype, const int64_t* pos,
                      const int64_t* neg, const float* d_pos, const float* d_neg, const float* w, float eps,
                      const float* grad_scale_dev, float* grad_z, segger_stream_t stream);

size_t segger_triplet_workspace_bytes(int64_t n_edges);
int segger_triplet_fwd(const segger_triplet_args* args, segger_stream_t stream);
int segger_triplet_bwd(const segger_triplet_args* args, segger_stream_t stream);

/*
 * segger_loss_head_fwd / _bwd: LitISTEncoder.get_losses after the sampling (lightning_model.py:151-213: loss_tx =
 * TripletLoss, loss_bd = MetricLoss (models/triplet_loss.py:128-204), loss_sg = TripletMarginLoss or BCEWithLogits on the
 * tx-belongs-bd edges (:167-207), weighted sum :210-211) as ONE launch each way (csrc/loss_head.hip).  Same arithmetic as
 * segger_triplet_* / segger_metric_* / segger_loss_combine_*, which remain for other shapes.
 *   loss_tx triplets are (t, tx_pos[t], tx_neg[t]), t < n_tx (rows of z_tx; tx_pos[t] < 0 skips t; mean over n_tx);
 *   loss_bd as segger_metric_fwd over the rows of z_bd;  loss_sg over n_sg triplets (sg_src -> z_tx, sg_pos / sg_neg -> z_bd;
 *   sg_pos < 0 skips; mean over n_sg, or over 2 n_sg logits for SEGGER_LOSS_BCE).
 *   out[0..2] = a[i] * loss_i, out[3] = sum_i b[i] out[i]  (a, b: float[3] on the device, as segger_loss_combine_fwd).
 * Forward, with tx_w / tx_state / tx_next / tx_hot_id / tx_hot_acc given (a training step): per loss_tx triplet the reciprocal
 *   distances of ACTIVE triplets (tx_w [n_tx, 2], 0 = inactive) and the chains of contributions per target row: tx_state
 *   int32 [2 n_tx + 4] = per row (chain head, contribution count), then the hot-row counter -- the CALLER ZERO-FILLS it
 *   before every forward launch (a chain walk is bounded whatever it holds: a stale buffer gives wrong numbers, never a
 *   spinning kernel); tx_next int32 [2 n_tx]; a row that receives more than 64 contributions (skewed draws: few members of a
 *   much-drawn cluster in the batch) is "hot": tx_hot_id int32 [n_tx + segger_loss_head_max_hot_rows(n_tx)] and tx_hot_acc fp32 [segger_loss_head_max_hot_rows(n_tx),
 *   C] hold its accumulator (no initialisation needed).  With grad_bd given the launch zero-fills it; with grad_out
 *   (float[4], d / d out) given it also writes grad_raw[3], the scale factors of the three backward terms
 *   (segger_loss_combine_bwd's result).
 * Backward: grad_tx [n_tx, C] (dtype of the embeddings) is WRITTEN, every row once, no atomics: row r = its anchor term of
 *   loss_tx + the positive / negative terms of the triplets chained to r + the anchor term of the segmentation triplet
 *   sg_of_tx[r] (int32 [n_tx], -1 = none: requires that no transcript anchors two segmentation triplets; NULL = the
 *   segmentation loss contributes nothing to grad_tx, the caller adds it); with y_tx given (z_tx = y_tx / max(|y_tx|,
 *   norm_eps) row-wise) the row is pushed through that normalisation's backward and grad_tx is d / d y_tx.  Hot rows are
 *   accumulated with fp32 atomics by the groups of the contributing triplets and finished by the last contributor.
 *   grad_bd [n_bd, C] fp32 is ACCUMULATED INTO (zero-filled by the forward): metric loss, segmentation positives (summed
 *   per boundary over sg_pos_indptr [n_bd + 1] / sg_pos_eid [n_sg], the triplets grouped by positive row) and negatives.
 * workspace: segger_loss_head_workspace_bytes() (per-block partial sums; the forward is TWO launches: the loss kernel and a
 *   one-workgroup launch that adds the partial sums in a fixed order -- a last-block-by-ticket form paid a whole-L2
 *   write-back per block for its device-scope release: 0.76 of 0.90 ms at 10^6 rows); ticket: unused (kept for layout).
 *   C in {32, 64, 128}.  Worth it below ~10^5 transcript rows (a captured 1M-edge step); at 10^6 rows the forward's chain
 *   atomics (4 returning atomics per triplet) make it slower than the kernel-by-kernel head (profiles/r04_loss_head_modes_c2.txt).
 */
#define SEGGER_LOSS_HEAD_DEFER_FINISH 1   /* in reserved_: with grad_out given, the forward leaves its partial sums and the
                                             backward launch (one extra workgroup) finishes the losses into `out`; both calls
                                             get the same workspace, out and grad_out: a captured training step's form */
typedef struct segger_loss_head_args {
  const void* z_tx; int64_t ld_ztx; int64_t n_tx;
  const void* z_bd; int64_t ld_zbd; int64_t n_bd;
  int32_t channels, dtype;
  const int64_t* tx_pos; const int64_t* tx_neg; float tx_margin, tx_eps;
  const int64_t* bd_pos; const int64_t* bd_neg; const float* bd_dpos; const float* bd_dneg; const float* bd_w; float bd_eps;
  int32_t sg_kind;
  const int64_t* sg_src; const int64_t* sg_pos; const int64_t* sg_neg; int64_t n_sg; float sg_margin, sg_eps;
  const int64_t* sg_pos_indptr; const int32_t* sg_pos_eid; const int32_t* sg_of_tx;
  const float* a; const float* b; const float* grad_out; float* out; float* grad_raw;
  float* tx_w; int32_t* tx_state; int32_t* tx_next; int32_t* tx_hot_id; float* tx_hot_acc;
  const void* y_tx; int64_t ld_ytx; float norm_eps; int32_t reserved_;
  void* grad_tx; int64_t ld_gtx; float* grad_bd;
  void* workspace; size_t workspace_bytes; int32_t* ticket;
} segger_loss_head_args;
size_t segger_loss_head_workspace_bytes(int64_t n_tx, int64_t n_bd, int64_t n_sg);
int64_t segger_loss_head_max_hot_rows(int64_t n_tx);
int segger_loss_head_supported(int32_t channels, int32_t dtype);
int segger_loss_head_fwd(const segger_loss_head_args* args, segger_stream_t stream);
int segger_loss_head_bwd(const segger_loss_head_args* args, segger_stream_t stream);

/* ------------------------------------------------------------------------
 * Positional embedder, stage 1: per-graph min / max of node positions.
 * Replaces the Python loop over graphs in Positional2dEmbedder.forward
 * (src/segger/models/ist_encoder.py:66-73: one boolean mask, two reductions and a
 * host sync per graph).
 *   pos   : [n, 2] fp32 ;  batch : [n] int64 graph id per node, or NULL (one graph)
 *   mins / maxs : [n_graphs, 2] fp32 ; a graph without nodes gets (0, 0) as in the
 *   reference; ids outside [0, n_graphs) are ignored.
 * segger_segment_minmax_ex, flags: SEGGER_MINMAX_INITIALISED = the caller has already filled mins with +inf and maxs with
 *   -inf (e.g. in its staging launch); SEGGER_MINMAX_KEEP_EMPTY = leave (+inf, -inf) for graphs without nodes (a consumer
 *   that only looks up the graphs of existing nodes never reads them): one launch instead of three.
 * ---------------------------------------------------------------------- */
#define SEGGER_MINMAX_INITIALISED 1
#define SEGGER_MINMAX_KEEP_EMPTY 2
int segger_segment_minmax_ex(const float* pos, const int64_t* batch, int64_t n, int64_t n_graphs, float* mins, float* maxs,
                             int32_t flags, segger_stream_t stream);
int segger_segment_minmax(const float* pos, const int64_t* batch, int64_t n, int64_t n_graphs,
                          float* mins, float* maxs, segger_stream_t stream);

/* ------------------------------------------------------------------------
 * Tall-skinny projection on the matrix cores (v_mfma_f32_32x32x16):
 *     y[n, m_out] = x[n, k_in] * w[m_out, k_in]^T (+ bias)
 * Replaces the cuBLAS GEMMs behind PyG's Linear for GATv2Conv.lin_l / lin_r
 * (constructed at src/segger/models/ist_encoder.py:111-124), HeteroDictLinear
 * (ist_encoder.py:282-286,328) and, called with w = W^T, their data gradients.
 * Every workgroup owns 128 rows and ALL m_out columns, so x is read from HBM once.  From ~2*10^5 rows on, for
 * k_in in {64, 128} and m_out in {64, 128, 384}, one persistent workgroup per CU keeps the whole w in LDS and its waves
 * walk 32-row tiles independently (same arithmetic, bit-identical results).
 *   dtype: SEGGER_BF16 / SEGGER_F16 (x, w, y); bias fp32 or NULL; fp32 accumulation.
 *   k_in in {64, 128, 256, 384}; m_out a multiple of 64; w contiguous [m_out, k_in].
 * segger_linear_supported() tells the host whether a shape is covered (others go to
 * the vendor GEMM library).
 * ---------------------------------------------------------------------- */
int segger_linear_supported(int32_t k_in, int32_t m_out, int32_t dtype);
int segger_linear_fwd(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy,
                      int64_t n_rows, int32_t k_in, int32_t m_out, int32_t dtype, segger_stream_t stream);
/* segger_linear_fwd_silu_grad: y = (x @ w^T) * silu'(gate), gate [n_rows, m_out] in `dtype`: the data gradient through
 * Linear -> SiLU (the positional MLP, ist_encoder.py:45-49) with torch's separate silu_backward pass folded into the
 * GEMM epilogue.  k_in = 64. */
int segger_linear_fwd_silu_grad(const void* x, int64_t ldx, const void* w, const void* gate, int64_t ld_gate, void* y,
                                int64_t ldy, int64_t n_rows, int32_t k_in, int32_t m_out, int32_t dtype,
                                segger_stream_t stream);
/* segger_linear_fwd_rowbias: the same with  y[row, :] += rowbias[rowidx[row], :]  added in the epilogue (table
 * [n_ids, ld_rb >= m_out] in `dtype`, 16-byte aligned rows, added in fp32; int32 ids in range -- not checked): the gene-embedding half of the first layer's projections
 * collapses to a per-gene table T = gelu(E) Wa^T + b ([n_genes, 384]), so the GEMM runs over the positional half only
 * (K 256 -> 128) and gelu(cat(E[g], pe)) is never materialised (ist_encoder.py:312-325 + GATv2Conv.lin_l / lin_r). */
int segger_linear_fwd_rowbias(const void* x, int64_t ldx, const void* w, const float* bias, const void* rowbias,
                              int64_t ld_rb, const int32_t* rowidx, void* y, int64_t ldy, int64_t n_rows, int32_t k_in,
                              int32_t m_out, int32_t dtype, segger_stream_t stream);

/* segger_linear_wgrad_f32_split: segger_linear_wgrad for fp32 storage on the bf16 matrix pipe (both operands split three ways,
 * six partial products; csrc/linear_f32_split.hip) -- the weight gradient autograd forms as `grad.t() @ x` for the nn.Linear
 * maps of ist_encoder.py:111-124,282-286.  Same arguments, workspace (segger_linear_wgrad_workspace_bytes) and deterministic
 * slab-order sums as segger_linear_wgrad; error within the exact-fp32 kernel's own, not bit-identical to it. */
/* segger_linear_fwd_f32_act: y = x @ W^T + b AND y_act = act(y) (act_kind 1 = GELU, 2 = SiLU) from one kernel at fp32 storage:
 * the pre-activation the backward keeps and the activation the next layer reads (the positional MLP's Linear -> SiLU and its
 * output's GELU, ist_encoder.py:43-49,320) without torch's separate elementwise pass.  Exact-fp32 kernel, k_in 64 / 128 / 256. */
int segger_linear_fwd_f32_act(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                              float* y_act, int64_t ld_yact, int32_t act_kind, int64_t n_rows, int32_t k_in, int32_t m_out,
                              segger_stream_t stream);
/* segger_linear_fwd_f32_gate: y = (x @ W^T) * act'(gate) at fp32 storage -- the data gradient through GELU (gate_kind 1) or
 * SiLU (2) with torch's separate gelu_backward / silu_backward pass folded into the GEMM epilogue (ist_encoder.py:47,320:
 * the positional MLP's SiLU, the GELU on the first layer's input).  w_is_planes != 0: W as the three bf16 planes of
 * segger_f32_split_planes (shapes of segger_linear_fwd_f32_split_supported); else W [m_out, k_in] fp32 on the exact-fp32
 * kernel (k_in 64 / 128 / 256).  gate [n_rows, ld_gate >= m_out] fp32. */
int segger_linear_fwd_f32_gate(const float* x, int64_t ldx, const void* w, int32_t w_is_planes, const float* gate,
                               int64_t ld_gate, int32_t gate_kind, float* y, int64_t ldy, int64_t n_rows, int32_t k_in,
                               int32_t m_out, segger_stream_t stream);
/* segger_f32_split_planes: planes [3][rows * cols] bf16 (hi, mid, lo: they add up to the fp32 number exactly) of w [rows, cols]
 * fp32 row-major, laid out as w or, transpose != 0, as w^T [cols, rows]: the weight operand of segger_linear_fwd_f32_split,
 * refreshed after every optimizer step. */
int segger_f32_split_planes(const float* w, int32_t rows, int32_t cols, int32_t transpose, void* planes, segger_stream_t stream);
/* ... for up to SEGGER_PLANES_MAX_JOBS matrices in ONE launch (every fp32 weight pack of a model after an optimizer step:
 * nn.Linear / GATv2Conv.lin_l / lin_r weights of ist_encoder.py:111-124,261,282-286); per job the arguments of
 * segger_f32_split_planes. */
#define SEGGER_PLANES_MAX_JOBS 32
typedef struct segger_planes_job {
  const float* w;      /* [rows, cols] fp32, contiguous */
  int32_t rows, cols;
  int32_t transpose;   /* non-zero: planes of w^T */
  int32_t reserved_;
  void* planes;        /* bf16 [3][rows * cols] */
} segger_planes_job;
int segger_f32_split_planes_many(const segger_planes_job* jobs, int32_t n_jobs, segger_stream_t stream);
int segger_linear_wgrad_f32_split_supported(int32_t m_out, int32_t k_in);
int segger_linear_wgrad_f32_split(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t n_rows, int32_t m_out,
                                  int32_t k_in, float* grad_w, float* grad_b, void* workspace, size_t workspace_bytes,
                                  segger_stream_t stream);

/* segger_gene_table_fwd/bwd: the per-gene table of segger_linear_fwd_rowbias and everything that flows back through it,
 * one launch each way (csrc/gene_table.hip) -- torch formed it with ~25 small launches per step (gelu, cat, vendor GEMMs,
 * slice gradients).  The first hetero layer projects gelu(cat(E[gene], pe)) with n_w (<= 4) stacked nn.Linear maps
 * w[i] [m[i], ld_w[i] >= 2 D] fp32 (GATv2Conv.lin_l / lin_r, ist_encoder.py:111-124), bias b[i] [m[i]] or NULL; M = sum m[i].
 *   fwd: tab [n_genes, ld_tab >= M] = gelu(E) Wa^T + b, Wa = w[:, 0:D], in `dtype`; wc [M, D] = w[:, D:2D] and wc_t [D, M]
 *        its transpose, both in `dtype` (either may be NULL).
 *   bwd: from g_tab [n_genes, M] fp32 (rows of dY summed by gene) and g_wc [M, D] fp32 (dWc of the positional GEMM; NULL:
 *        right halves left untouched): g_table [n_genes, D] = gelu'(E) * (g_tab Wa) (NULL to skip); g_w[i] [m[i], 2 D]
 *        contiguous fp32 = [g_tab^T gelu(E) | g_wc] rows of weight i; g_b[i] [m[i]] = column sums of g_tab (NULL to skip). */
typedef struct segger_gene_table_args {
  const float* table; int32_t n_genes; int32_t D; int32_t n_w; int32_t dtype;
  const float* w[4]; int64_t ld_w[4]; int32_t m[4]; const float* b[4];
  void* tab; int64_t ld_tab; void* wc; void* wc_t;                    /* forward outputs */
  const float* g_tab; const float* g_wc;                              /* backward inputs */
  float* g_table; float* g_w[4]; float* g_b[4];                       /* backward outputs */
} segger_gene_table_args;
int segger_gene_table_fwd(const segger_gene_table_args* args, segger_stream_t stream);
int segger_gene_table_bwd(const segger_gene_table_args* args, segger_stream_t stream);

/* segger_linear_fwd_f32_split: the fp32-storage projection on the bf16 matrix pipe -- every fp32 operand as the sum of three
 * bf16 numbers, the product as its six leading partial products, each exact in the MFMA's fp32 accumulator
 * (csrc/linear_f32_split.hip).  An OPT-IN alternative to segger_linear_fwd's exact-fp32 kernel for the same nn.Linear call
 * sites (ist_encoder.py:111-124,282-286 under a default fp32 Trainer, cli/segment.py:400-405): error ~2^-22 relative to
 * sum |x||w| instead of fp32 rounding only, at 6/16 of the bf16 MFMA rate instead of the 157 TFLOP/s fp32 pipe.
 *   x [n, k_in] fp32, y [n, m_out] fp32 (16-byte aligned rows); w3 = bf16 [3][m_out][k_in]: hi = bf16(w), mid = bf16(w - hi),
 *   lo = bf16(w - hi - mid).  Covered: k_in 128 with m_out a multiple of 64; k_in 384 with m_out 128 (dX = dY W on W^T planes).
 * RANGE: not range-equivalent to the exact kernel.  bf16 has fp32's exponent range but rounds up at the top: an operand with
 * |v| > 0x7F7F8000 as fp32 bits (3.3895e38: within 2^-9 of FLT_MAX) rounds its hi plane to +-inf and the result is NaN
 * (inf - inf in the mid plane) where the exact kernel stays finite; sub-normal operands lose the planes below 2^-133.  Node
 * features behind a GELU / L2 normalisation are nowhere near either end; callers that cannot promise it select the exact
 * kernels (host side: SEGGER_AMD_F32_EXACT=1 / ops.F32_SPLIT = False -- the active mode is recorded on bench.py's line,
 * f32.projections). */
int segger_linear_fwd_f32_split_supported(int32_t k_in, int32_t m_out);
int segger_linear_fwd_f32_split(const float* x, int64_t ldx, const void* w3, const float* bias, float* y, int64_t ldy,
                                int64_t n_rows, int32_t k_in, int32_t m_out, segger_stream_t stream);
/* ... with y[row, :] += rowbias[rowidx[row], :] in the epilogue (fp32 table [n_ids, ld_rb >= m_out]; no bias): the fp32 form
 * of segger_linear_fwd_rowbias on the split kernel. */
int segger_linear_fwd_f32_split_rowbias(const float* x, int64_t ldx, const void* w3, const float* rowbias, int64_t ld_rb,
                                        const int32_t* rowidx, float* y, int64_t ldy, int64_t n_rows, int32_t k_in,
                                        int32_t m_out, segger_stream_t stream);

/* segger_linear_fwd_pair: two such projections with the same k_in and dtype as ONE launch -- a hetero layer projects its
 * transcripts ([lin_l | lin_r | lin_l], ist_encoder.py:109-134 through HeteroConv) and its ~10^2-10^3 boundaries (lin_r)
 * in the same step, and lin_last maps both node types (ist_encoder.py:282-286,328); the small one's blocks ride in the
 * large one's grid instead of paying a launch of their own (a large `a` takes its persistent resident-W launch and `b` its
 * own; fp32: two launches).  Same arithmetic as segger_linear_fwd, bit-identical results. */
typedef struct segger_linear_args {
  const void* x; int64_t ldx; const void* w; const float* bias; void* y; int64_t ldy; int64_t n_rows; int32_t m_out;
  int32_t reserved_;
} segger_linear_args;
int segger_linear_fwd_pair(const segger_linear_args* a, const segger_linear_args* b, int32_t k_in, int32_t dtype,
                           segger_stream_t stream);
/* ... and with an input width per side (a first layer's two data gradients dX = dY W read 384 and 128 columns of dY): one
 * launch for (k_a, k_b) = (384, 128) or equal widths, two otherwise. */
int segger_linear_fwd_pair_k(const segger_linear_args* a, int32_t k_a, const segger_linear_args* b, int32_t k_b,
                             int32_t dtype, segger_stream_t stream);

/*
 * segger_linear_wgrad: the parameter gradients of the same projections,
 *     grad_w[M, K] = dY[n, M]^T * X[n, K]      grad_b[M] = sum_n dY[n, :]       (fp32 outputs)
 * i.e. what torch autograd computes for nn.Linear / PyG Linear (lin_l, lin_r, lin_last, the positional MLP;
 * src/segger/models/ist_encoder.py:44-48,111-124,261,282-286) as `grad.t() @ x` and `grad.sum(0)`.
 * One pass over dY and X on the matrix cores: row slabs per workgroup, the whole [M, K] accumulator in
 * registers, transposed operand reads from LDS (ds_read_b64_tr_b16), deterministic slab-order reduction.
 *   dy [n_rows, m_out] row stride ld_dy, x [n_rows, k_in] row stride ld_x (elements; 16-byte aligned rows)
 *   covered: m_out in {64,128,192,384}, k_in in {64,128,256}, bf16 / f16 (segger_linear_wgrad_supported)
 *   grad_b may be NULL; workspace >= segger_linear_wgrad_workspace_bytes(n_rows, m_out, k_in)
 */
int segger_linear_wgrad_supported(int32_t m_out, int32_t k_in, int32_t dtype);
size_t segger_linear_wgrad_workspace_bytes(int64_t n_rows, int32_t m_out, int32_t k_in);
int segger_linear_wgrad(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, int64_t n_rows,
                        int32_t m_out, int32_t k_in, int32_t dtype, float* grad_w, float* grad_b,
                        void* workspace, size_t workspace_bytes, segger_stream_t stream);
/* segger_linear_wgrad_dx: the WHOLE backward of one projection in one pass over dY:
 *     grad_w, grad_b as above, and    dx[n, K] = dY[n, M] * W[M, K]
 * (autograd's `grad @ weight` for the same nn.Linear / PyG Linear call sites).  dY [n, 3*HC] of the stacked
 * lin_l | lin_r | lin_l projections is the widest matrix of a layer's backward; the separate data-gradient GEMM
 * (segger_linear_fwd on W^T) read it a second time.  w_t = W^T, [k_in, m_out] contiguous in `dtype`; dx row stride
 * ld_dx (elements, 16-byte aligned rows); covered: k_in == 128, m_out in {64,128,192,384}, bf16 / f16
 * (segger_linear_wgrad_dx_supported); workspace as for segger_linear_wgrad.
 * gelu_gate (optional, [n, k_in] in `dtype`, row stride ld_gate): dx[row, c] *= gelu'(gelu_gate[row, c]) in the epilogue
 * -- the projection's input was gelu(gate) (ISTEncoder's GELU on the concatenated first-layer input,
 * ist_encoder.py:320): the gradient leaves as d / d gate without an elementwise pass of its own; covered for m_out in
 * {128, 384} (segger_linear_wgrad_dx_gate_supported); NULL = plain dX. */
int segger_linear_wgrad_dx_supported(int32_t m_out, int32_t k_in, int32_t dtype);
int segger_linear_wgrad_dx_gate_supported(int32_t m_out, int32_t k_in, int32_t dtype);
int segger_linear_wgrad_dx(const void* dy, int64_t ld_dy, const void* x, int64_t ld_x, const void* w_t, int64_t n_rows,
                           int32_t m_out, int32_t k_in, int32_t dtype, float* grad_w, float* grad_b, void* dx,
                           int64_t ld_dx, const void* gelu_gate, int64_t ld_gate, void* workspace, size_t workspace_bytes,
                           segger_stream_t stream);
/* segger_linear_wgrad_pair: the backward passes of TWO projections with the same k_in and dtype as one launch -- what
 * segger_linear_fwd_pair is to the forward: a hetero layer's transcript-side and boundary-side projections, lin_last of both
 * node types; the small side's one or two workgroups ride in the large side's grid.  With w_t / dx given on both sides the
 * one-pass form (segger_linear_wgrad_dx), with both NULL the weight / bias gradients only (segger_linear_wgrad).  Same
 * arithmetic and workspaces as the single calls (bit-identical results); shape pairs without a paired kernel, fp32 and
 * empty sides run as two launches. */
typedef struct segger_wgrad_args {
  const void* dy; int64_t ld_dy; const void* x; int64_t ld_x; const void* w_t; int64_t n_rows; int32_t m_out; int32_t reserved_;
  float* grad_w; float* grad_b; void* dx; int64_t ld_dx; void* workspace; size_t workspace_bytes;
} segger_wgrad_args;
int segger_linear_wgrad_pair(const segger_wgrad_args* a, const segger_wgrad_args* b, int32_t k_in, int32_t dtype,
                             segger_stream_t stream);

/*
 * Deferred partial sums.  segger_linear_wgrad / _wgrad_dx / segger_posmlp_wgrad and segger_gatv2_bwd finish with a small
 * kernel that sums per-workgroup partials from their workspace into grad_w / grad_b / grad_att / grad_bias.  Between
 *     segger_reductions_defer_begin()  ...  segger_reductions_flush(stream)
 * (process-wide: autograd queues from its own device thread) those sums are queued instead of launched and the flush runs them
 * all as ONE grid (two when a sum has more than 128 partials: those fold into 32 groups first) -- autograd hands every parameter gradient of a backward pass (ist_encoder.py:289-333 under
 * lightning_model.py:215-237) to the optimizer at the same time anyway.  Contract while deferring: the outputs are
 * undefined and the workspaces must stay untouched until the flush has been enqueued on the same stream (or one
 * ordered after the producers); at most 56 sums are queued, further ones run immediately.  The table travels as a
 * kernel argument: no allocation, no synchronisation, capturable.  segger_reductions_pending(): queued sums, -1 when
 * not deferring.
 */
int segger_reductions_defer_begin(void);
int segger_reductions_pending(void);
int segger_reductions_flush(segger_stream_t stream);

/* ------------------------------------------------------------------------
 * Encoder front end / tail as fused row-wise kernels.
 *
 * segger_posfreq: Positional2dEmbedder up to the MLP input
 *   (src/segger/models/ist_encoder.py:74 normalisation, :22-31,:51-55 sinusoid):
 *   out[(n*2 + c), :] = [cos(p f_j) | sin(p f_j)], p = (pos[n,c] - mins[g,c]) / (maxs[g,c] - mins[g,c] + eps),
 *   f_j = exp(-ln(max_period) * j / (freq_dim/2)), g = batch[n] (0 when batch is NULL); out is [2n, freq_dim] in `dtype`.
 *
 * segger_embed_gelu_fwd/bwd: x0 = gelu(cat(table[ids], pe)) (ist_encoder.py:312-320 for the 'tx' type):
 *   table fp32 [n_rows_table, D] (nn.Embedding weight), ids int32 [n], pe [n, D] -> x0 [n, 2D].
 *   bwd: gpe = gx0[:, D:] * gelu'(pe); gtable (optional, NULL to skip) = gelu'(table) * sum over the rows of each
 *   id of gx0[:, :D].  The rows come GROUPED BY ID: gene_ptr int64 [n_rows_table + 1] / gene_rows int32 [n] are the
 *   indptr / col arrays of segger_csr_from_coo(row = ids, col = 0..n-1) (one sort per batch); the sum is a
 *   deterministic two-stage segmented reduction (no atomics).  D % 8 == 0, D <= 2048.
 *
 * segger_l2norm_fwd/bwd: torch.nn.functional.normalize(dim=-1, eps) (ist_encoder.py:331-332) and its gradient;
 *   channels in {8, 16, 32, 64, 128}.
 *
 * segger_colsum: out[c] = sum_n x[n, c] in fp32 -- the bias gradient of every nn.Linear / PyG Linear on the path
 *   (lin_l / lin_r of GATv2Conv, ist_encoder.py:111-124; pos-embedding MLP :45-49; lin_last :282-286), which autograd
 *   computes as grad_output.sum(0).  cols % 8 == 0, cols <= 2048; deterministic (two-stage, no atomics).
 * ---------------------------------------------------------------------- */
int segger_posfreq(const float* pos, const int64_t* batch, const float* mins, const float* maxs, int64_t n,
                   int32_t freq_dim, float eps, float max_period, void* out, int32_t dtype, segger_stream_t stream);
/*
 * segger_posmlp_fwd: the whole Positional2dEmbedder (ist_encoder.py:33-79) in one kernel -- normalisation, the
 * 256-wide sinusoid, Linear(256 -> 64), SiLU, Linear(64 -> 64) -- with the sinusoid features generated in registers as
 * MFMA operand fragments instead of written to / read from HBM (segger_posfreq + two segger_linear_fwd otherwise).
 *   w0 [64, 256], w2 [64, 64] row-major in `dtype` (bf16 / f16), b0 / b2 fp32 [64]
 *   pe   [2n, 64] in `dtype` (= the embedder's [n, 128] output, x half then y half per node)
 *   z1, h1 [2n, 64] in `dtype`, pn [2n] fp32: the pre-activation of the first layer, its SiLU and the normalised
 *        coordinate per row, stored for the backward; all NULL for inference; h1 may be NULL on its own
 *        (segger_posmlp_bwd recomputes it from z1).
 *   gelu != 0: pe receives gelu(embedder output) -- the positional half of ISTEncoder's gelu(cat(...)),
 *        ist_encoder.py:324-325 -- and, when training, pe_pre [2n, 64] the output itself (for gelu'); else pe_pre NULL.
 * segger_posmlp_wgrad: dW0 [64, 256] = dz1^T F and db0 [64] = sum dz1 with F regenerated from pn inside the weight-
 *   gradient kernel (one float per row read instead of a 512-byte feature row); workspace as
 *   segger_linear_wgrad_workspace_bytes(n_rows, 64, 256); n_rows = 2n.
 * segger_posmlp_bwd: the embedder's whole backward from ONE read of g = d loss / d pe ([n_rows = 2n, 64], row stride
 *   ld_g): dW2 = g^T SiLU(z1), db2 = sum g, dz1 = (g W2) * SiLU'(z1) (kept on the chip), dW0 = dz1^T F(pn), db0 =
 *   sum dz1 -- autograd's two `grad.t() @ x`, two `grad.sum(0)`, the `grad @ W` and the SiLU backward of
 *   ist_encoder.py:45-49, 76-79.  w2_t = W2^T [64, 64] row-major in `dtype`; outputs fp32; workspace
 *   segger_posmlp_bwd_workspace_bytes(n_rows); deterministic (per-workgroup partials summed in order).
 * segger_posmlp_bwd_pair: the same for TWO row sets in one launch and one sum of partials -- ISTEncoder embeds the
 *   transcripts' and the boundaries' positions with the same Positional2dEmbedder (reference ist_encoder.py:314-318), so
 *   its four parameters receive ONE gradient (autograd would add the two calls' gradients with four more launches).
 *   Either side may be empty (n_rows 0, pointers ignored); workspace segger_posmlp_bwd_pair_workspace_bytes(n_a, n_b).
 * Covered: frequency_embedding_size 256, hidden_size 128 (segger's in_channels default): segger_posmlp_supported.
 */
int segger_posmlp_supported(int32_t freq_dim, int32_t dim, int32_t dtype);
/*
 * segger_posenc_poly_*: the first Linear of Positional2dEmbedder (ist_encoder.py:45-49 `mlp[0]` applied to the sinusoid of
 * :22-31) at fp32 storage WITHOUT the [2n, freq_dim] feature matrix and without a GEMM.  The embedder's argument is a min-max
 * NORMALISED coordinate p in [0, 1] (:62-64, :74) and every frequency is <= 1, so |w_j p| <= 1: cos / sin equal their Taylor
 * series through p^12 to 1/13! = 1.6e-10, and the sum over the features can be taken first:
 *   z1[row, m] = sum_{d=0..12} coef[m, d] p^d,   coef[m, d] = [d = 0] b0[m] + (-1)^floor(d/2)/d! sum_j W0[m, (d odd ? half : 0) + j] w_j^d
 * segger_posenc_poly_coef: coef [dim][16] fp32 (13 terms + padding) from W0 [dim, freq_dim] / b0 (NULL: none), in float64;
 *   once per step.
 * segger_posenc_poly_fwd: z1 [2n, dim] fp32 (row = 2 * node + axis), optionally h1 = SiLU(z1) and pn [2n] (the normalised
 *   coordinate, all the weight gradient needs); exact fp32 FMA arithmetic.
 * segger_posenc_poly_wgrad: dW0 [dim, freq_dim] = dz1^T F and db0 [dim] = sum dz1 as  M V^T  with the 13 moments
 *   M[m, d] = sum_rows dz1[row, m] pn[row]^d (fp32 per lane, float64 across workgroups, fixed order: deterministic) and V
 *   the features' Taylor coefficients; dz1 [n_rows, dim] fp32 (row stride ld); workspace
 *   segger_posenc_poly_wgrad_workspace_bytes(n_rows, dim).
 * Covered: dim 32 / 64 / 128, even freq_dim <= 4096 (segger: 64, 256), max_period > 1: segger_posenc_poly_supported.
 */
int segger_posenc_poly_supported(int32_t freq_dim, int32_t dim);
int segger_posenc_poly_coef(const float* w0, const float* b0, int32_t freq_dim, int32_t dim, float max_period, float* coef,
                            segger_stream_t stream);
int segger_posenc_poly_fwd(const float* pos, const int64_t* batch, const float* mins, const float* maxs, int64_t n, float eps,
                           const float* coef, int32_t dim, float* z1, float* h1, float* pn, segger_stream_t stream);
size_t segger_posenc_poly_wgrad_workspace_bytes(int64_t n_rows, int32_t dim);
int segger_posenc_poly_wgrad(const float* dz1, int64_t ld, const float* pn, int64_t n_rows, int32_t freq_dim, int32_t dim,
                             float max_period, float* grad_w0, float* grad_b0, void* workspace, size_t workspace_bytes,
                             segger_stream_t stream);
int segger_posmlp_fwd(const float* pos, const int64_t* batch, const float* mins, const float* maxs, int64_t n, float eps,
                      float max_period, const void* w0, const float* b0, const void* w2, const float* b2, void* pe,
                      void* z1, float* pn, void* h1, void* pe_pre, int32_t gelu, int32_t dtype, segger_stream_t stream);
size_t segger_posmlp_bwd_workspace_bytes(int64_t n_rows);
int segger_posmlp_bwd(const void* g, int64_t ld_g, const void* z1, const float* pn, const void* w2_t, int64_t n_rows,
                      float max_period, int32_t dtype, float* grad_w0, float* grad_b0, float* grad_w2, float* grad_b2,
                      void* workspace, size_t workspace_bytes, segger_stream_t stream);
size_t segger_posmlp_bwd_pair_workspace_bytes(int64_t n_rows_a, int64_t n_rows_b);
int segger_posmlp_bwd_pair(const void* g_a, int64_t ld_ga, const void* z1_a, const float* pn_a, int64_t n_rows_a,
                           const void* g_b, int64_t ld_gb, const void* z1_b, const float* pn_b, int64_t n_rows_b,
                           const void* w2_t, float max_period, int32_t dtype, float* grad_w0, float* grad_b0,
                           float* grad_w2, float* grad_b2, void* workspace, size_t workspace_bytes,
                           segger_stream_t stream);
int segger_posmlp_wgrad(const void* dz1, int64_t ld_dz1, const float* pn, int64_t n_rows, float max_period,
                        int32_t dtype, float* grad_w0, float* grad_b0, void* workspace, size_t workspace_bytes,
                        segger_stream_t stream);
int segger_embed_gelu_fwd(const float* table, const int32_t* ids, const void* pe, int64_t ld_pe, int64_t n,
                          int32_t n_rows_table, int32_t D, void* out, int64_t ld_out, int32_t dtype,
                          segger_stream_t stream);
size_t segger_embed_gelu_bwd_workspace_bytes(int64_t n, int32_t n_rows_table, int32_t D);
int segger_embed_gelu_bwd(const void* gx0, int64_t ld_g, const float* table, const void* pe, int64_t ld_pe,
                          int64_t n, int32_t n_rows_table, int32_t D, void* gpe, int64_t ld_gpe, float* gtable,
                          const int64_t* gene_ptr, const int32_t* gene_rows, void* workspace, size_t workspace_bytes,
                          int32_t dtype, segger_stream_t stream);
/* segger_front_join_fwd/bwd: the encoder's layer-0 input of BOTH node types, one launch each way
 * (ist_encoder.py:312-320: x = {k: gelu(cat(lin_first[k](x[k]), pos_emb(pos[k])))}):
 *   out_tx[r] = gelu(cat(table[ids[r]], pe[r]))  r < n_tx;   out_bd[r] = gelu(cat(xb[r], pe[n_tx + r]))  r < n_bd
 * with `pe` the positional embeddings of the two types back to back ([n_tx + n_bd, D]: one embedder call) and xb =
 * lin_first['bd'](x_bd).  bwd: g_pe [n_tx + n_bd, D] = g[:, D:] * gelu'(pe) written as ONE matrix, g_xb = g_bd[:, :D] *
 * gelu'(xb), g_table (optional) as segger_embed_gelu_bwd (gene_ptr / gene_rows / workspace of
 * segger_embed_gelu_bwd_workspace_bytes(n_tx, n_rows_table, D)).  It replaces torch's cat + GELU (+ their backward, and
 * the full-size cat that joins the two slices' gradients) on the boundary side of a small batch. */
typedef struct segger_front_join_args {
  const float* table; const int32_t* ids; int32_t n_rows_table; int32_t D; int32_t dtype; int32_t reserved_;
  const void* pe; int64_t ld_pe; int64_t n_tx, n_bd;
  const void* xb; int64_t ld_xb;
  void* out_tx; int64_t ld_out_tx; void* out_bd; int64_t ld_out_bd;                      /* forward outputs [n, 2D] */
  const void* g_tx; int64_t ld_g_tx; const void* g_bd; int64_t ld_g_bd;                  /* backward inputs [n, 2D] */
  void* g_pe; int64_t ld_g_pe; void* g_xb; int64_t ld_g_xb;                              /* backward outputs */
  float* g_table; const int64_t* gene_ptr; const int32_t* gene_rows; void* workspace; size_t workspace_bytes;
} segger_front_join_args;
int segger_front_join_fwd(const segger_front_join_args* args, segger_stream_t stream);
int segger_front_join_bwd(const segger_front_join_args* args, segger_stream_t stream);
int segger_l2norm_fwd(const void* y, int64_t ld_y, int64_t n, int32_t channels, float eps, void* z, int64_t ld_z,
                      int32_t dtype, segger_stream_t stream);
int segger_l2norm_bwd(const void* y, int64_t ld_y, const void* gz, int64_t ld_gz, int64_t n, int32_t channels,
                      float eps, void* gy, int64_t ld_gy, int32_t dtype, segger_stream_t stream);
/* segger_l2norm_many: up to 4 row normalisations in ONE launch (both node types of the encoder's tail, ist_encoder.py:331-332;
 * the boundary side's backward from the loss head's fp32 gradient).  Per segment: gz == NULL -> out = y / max(|y|, eps);
 * otherwise out = d / d y from the incoming gradient gz ([n, C] in the embeddings' dtype, or fp32 when gz_f32 != 0). */
typedef struct segger_l2norm_seg {
  const void* y; int64_t ld_y; int64_t n;
  void* out; int64_t ld_out;
  const void* gz; int64_t ld_gz; int32_t gz_f32; int32_t reserved_;
} segger_l2norm_seg;
int segger_l2norm_many(const segger_l2norm_seg* segs, int32_t n_segs, int32_t channels, float eps, int32_t dtype,
                       segger_stream_t stream);
/* the same with the incoming gradient given as the sum of two matrices gz + gz2 (gz2 may be NULL): the loss head hands
 * the anchors' stored rows and the atomically accumulated rows over separately (segger_triplet_args.grad_a_rows) */
int segger_l2norm_bwd2(const void* y, int64_t ld_y, const void* gz, int64_t ld_gz, const void* gz2, int64_t ld_gz2, int64_t n,
                       int32_t channels, float eps, void* gy, int64_t ld_gy, int32_t dtype, segger_stream_t stream);
/* out[s, :] = sum of the rows x[seg_rows[seg_ptr[s] : seg_ptr[s+1]], :] in fp32 (deterministic, no atomics): the
 * scatter-add of per-edge rows into a small table, e.g. the boundary-side gradient of the triplet loss
 * (lightning_model.py:182-187 autograd).  seg_ptr / seg_rows = indptr / col of segger_csr_from_coo(row = segment id of
 * every x row, col = 0..n-1).  D % 8 == 0, D <= 2048. */
size_t segger_segment_rowsum_workspace_bytes(int64_t n, int32_t n_segments, int32_t D);
int segger_segment_rowsum(const void* x, int64_t ld, int64_t n, int32_t D, int32_t dtype, const int64_t* seg_ptr,
                          const int32_t* seg_rows, int32_t n_segments, float* out, void* workspace,
                          size_t workspace_bytes, segger_stream_t stream);
size_t segger_colsum_workspace_bytes(int64_t n, int32_t cols);
int segger_colsum(const void* x, int64_t ld, int64_t n, int32_t cols, int32_t dtype, float* out,
                  void* workspace, size_t workspace_bytes, segger_stream_t stream);

/* ------------------------------------------------------------------------
 * Exact 2-D k-nearest neighbours on a uniform grid: the step BEFORE the hot path
 * (graph construction).  Replaces scipy.spatial.KDTree(points).query(queries, k,
 * distance_upper_bound=max_dist, workers=-1) in src/segger/data/utils/neighbors.py:122-163.
 *   points [n,2] fp32; queries [m,2] fp32 or NULL (= points, then m must equal n);
 *   grid: origin (x0,y0), square cells of side `cell`, nx x ny cells (points outside are
 *   clamped into the border cells, so any grid is valid; the host picks ~2 points per cell);
 *   nbr [m,k] int32: neighbour ids sorted by distance, padded with n (scipy's convention);
 *   dist [m,k] fp32 or NULL: distances, +inf for padding;  1 <= k <= 64; max_dist may be +inf.
 * ---------------------------------------------------------------------- */
size_t segger_knn_workspace_bytes(int64_t n_points, int32_t nx, int32_t ny);
int segger_knn_grid(const float* points, int64_t n_points, const float* queries, int64_t n_queries, int32_t k,
                    float max_dist, float x0, float y0, float cell, int32_t nx, int32_t ny,
                    int32_t* nbr, float* dist, void* workspace, size_t workspace_bytes, segger_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* SEGGER_AMD_H */
