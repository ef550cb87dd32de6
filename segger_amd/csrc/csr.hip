// segger_csr_from_coo: COO edge_index -> CSR (indptr, col, eid) by a stable
// device radix sort on the row id.  Built once per batch and edge type.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace segger {
namespace {

__global__ __launch_bounds__(256) void csr_prepare_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ colv,
                                                         int64_t n_edges, int64_t n_rows, int64_t n_cols,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                         int32_t* __restrict__ n_invalid) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  int64_t r = row[e];
  const int64_t c = colv[e];
  const bool bad = r < 0 || r >= n_rows || c < 0 || c >= n_cols;
  if (bad) {
    if (n_invalid) atomicAdd(n_invalid, 1);
    if (r < 0 || r >= n_rows) r = 0;
  }
  keys[e] = (uint32_t)r;
  vals[e] = (int32_t)e;
}

__global__ __launch_bounds__(256) void csr_finalize_kernel(const uint32_t* __restrict__ keys_sorted, const int32_t* __restrict__ eid,
                                                          const int64_t* __restrict__ colv, int64_t n_edges, int64_t n_rows,
                                                          int64_t n_cols, int64_t* __restrict__ indptr, int32_t* __restrict__ col) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_edges) return;
  int64_t c = colv[eid[s]];
  if (c < 0 || c >= n_cols) c = 0;
  col[s] = (int32_t)c;
  const int64_t k = keys_sorted[s];
  const int64_t kprev = s > 0 ? (int64_t)keys_sorted[s - 1] : -1;
  for (int64_t r = kprev + 1; r <= k; ++r) indptr[r] = s;      // first slot of row k; empty rows before it
  if (s == n_edges - 1)
    for (int64_t r = k + 1; r <= n_rows; ++r) indptr[r] = n_edges;
}

// one wave per window of W <= 64 rows: rank = number of rows of the window that come first
// (larger degree, ties by row id); order[base + rank] = row
__global__ __launch_bounds__(256) void csr_row_order_kernel(const int64_t* __restrict__ indptr, int64_t n_rows, int window,
                                                           int32_t* __restrict__ order) {
  const int lane = threadIdx.x & 63;
  const int per_wave = 64 / window;                       // windows handled by one wave
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t base = (wave_id * per_wave + lane / window) * window;   // first row of this lane's window
  const int wl = lane % window;
  const int64_t row = base + wl;
  int deg = -1;
  if (row < n_rows) deg = (int)(indptr[row + 1] - indptr[row]);
  int rank = 0;
  const int w0 = lane - wl;                               // first lane of the window
  for (int j = 0; j < window; ++j) {
    const int dj = __shfl(deg, w0 + j, 64);
    rank += (dj > deg) || (dj == deg && j < wl);
  }
  if (row < n_rows) order[base + rank] = (int32_t)row;    // rows past the end rank last: ranks of real rows stay < count
}

__global__ __launch_bounds__(256) void coo_unique_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_ids,
                                                        int32_t* __restrict__ marks, int32_t* __restrict__ unique) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int64_t i = ids[e];
  if (i < 0 || i >= n_ids) return;                       // reported by the CSR build's own validation
  if (atomicAdd(marks + i, 1) != 0) *unique = 0;
}

int key_bits(int64_t n_rows) {
  int b = 1;
  while (b < 32 && (1LL << b) < n_rows) ++b;
  return b;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp_bytes(int64_t n_edges, int64_t n_rows) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n_edges, 0, key_bits(n_rows), (hipStream_t)0);
  return bytes;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" size_t segger_csr_from_coo_workspace_bytes(int64_t n_edges, int64_t n_rows) {
  if (n_edges <= 0) return 256;
  return 3 * align_up((size_t)n_edges * 4) + align_up(sort_temp_bytes(n_edges, n_rows)) + 256;
}

extern "C" int segger_csr_from_coo(const int64_t* row, const int64_t* colv, int64_t n_edges, int64_t n_rows, int64_t n_cols,
                                   int64_t* indptr, int32_t* col, int32_t* eid, int32_t* n_invalid,
                                   void* workspace, size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_edges >= 0 && n_rows >= 0 && n_cols >= 0, "segger_csr_from_coo: negative size");
  SEGGER_REQUIRE(n_edges < 0x7fffffffLL && n_rows < 0x7fffffffLL && n_cols < 0x7fffffffLL,
                 "segger_csr_from_coo: more than 2^31-1 edges/rows/cols in one batch");
  SEGGER_REQUIRE(indptr != nullptr, "segger_csr_from_coo: indptr is NULL");
  if (n_invalid) SEGGER_HIP(hipMemsetAsync(n_invalid, 0, sizeof(int32_t), stream));
  if (n_edges == 0) {
    SEGGER_HIP(hipMemsetAsync(indptr, 0, (size_t)(n_rows + 1) * sizeof(int64_t), stream));
    return SEGGER_OK;
  }
  SEGGER_REQUIRE(row && colv && col && eid, "segger_csr_from_coo: NULL edge array");
  SEGGER_REQUIRE(n_rows > 0 && n_cols > 0, "segger_csr_from_coo: edges given but no rows/cols");
  const size_t need = segger_csr_from_coo_workspace_bytes(n_edges, n_rows);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("segger_csr_from_coo: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  const size_t seg = align_up((size_t)n_edges * 4);
  char* base = static_cast<char*>(workspace);
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(base);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + seg);
  int32_t* vals_in = reinterpret_cast<int32_t*>(base + 2 * seg);
  void* temp = base + 3 * seg;
  size_t temp_bytes = sort_temp_bytes(n_edges, n_rows);

  const unsigned nblk = (unsigned)((n_edges + 255) / 256);
  hipLaunchKernelGGL(csr_prepare_kernel, dim3(nblk), dim3(256), 0, stream, row, colv, n_edges, n_rows, n_cols,
                     keys_in, vals_in, n_invalid);
  SEGGER_LAUNCH_CHECK("csr_prepare_kernel");
  SEGGER_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, eid, (size_t)n_edges, 0,
                                       key_bits(n_rows), stream));
  hipLaunchKernelGGL(csr_finalize_kernel, dim3(nblk), dim3(256), 0, stream, keys_out, eid, colv, n_edges, n_rows, n_cols,
                     indptr, col);
  SEGGER_LAUNCH_CHECK("csr_finalize_kernel");
  return SEGGER_OK;
}

// ---- block tables: the distinct column ids of SEGGER_BLOCK_ROWS consecutive visiting positions ---------------------------
namespace segger { namespace {
constexpr int kTabSlots = 2048;                 // hash table (open addressing) for <= 1024 edges of a block
__global__ __launch_bounds__(256) void csr_block_tables_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                                              const int32_t* __restrict__ order, int64_t n_rows,
                                                              int32_t* __restrict__ blk_cnt, int32_t* __restrict__ blk_src,
                                                              uint8_t* __restrict__ col_local, int32_t* __restrict__ overflow) {
  __shared__ int32_t keys[kTabSlots];
  __shared__ int32_t ids[kTabSlots];
  __shared__ int64_t rbeg[SEGGER_BLOCK_ROWS];
  __shared__ int32_t rlen[SEGGER_BLOCK_ROWS];
  __shared__ int32_t counter, total;
  const int64_t blk = blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < kTabSlots; i += 256) keys[i] = -1;
  if (tid == 0) { counter = 0; total = 0; }
  __syncthreads();
  if (tid < SEGGER_BLOCK_ROWS) {
    const int64_t pos = blk * SEGGER_BLOCK_ROWS + tid;
    int64_t b = 0; int32_t len = 0;
    if (pos < n_rows) {
      const int64_t row = order ? (int64_t)order[pos] : pos;
      b = indptr[row]; len = (int32_t)(indptr[row + 1] - b);
    }
    rbeg[tid] = b; rlen[tid] = len;
    atomicAdd(&total, len);
  }
  __syncthreads();
  if (total > 1024) {                            // (the table could fill up: such a block keeps the plain gathers)
    if (tid == 0) { blk_cnt[blk] = -1; *overflow = 1; }
    return;
  }
  const int g = tid >> 4, l = tid & 15;          // 16 lanes per row
  auto slot_of = [](int32_t v) { return (int)(((uint32_t)v * 2654435761u) >> 21) & (kTabSlots - 1); };
  for (int e = l; e < rlen[g]; e += 16) {
    const int32_t v = col[rbeg[g] + e];
    int h = slot_of(v);
    while (true) {
      const int32_t old = atomicCAS(&keys[h], -1, v);
      if (old == -1 || old == v) break;
      h = (h + 1) & (kTabSlots - 1);
    }
  }
  __syncthreads();
  for (int sl = tid; sl < kTabSlots; sl += 256) {
    if (keys[sl] != -1) {
      const int id = atomicAdd(&counter, 1);     // (which slot number a row gets varies from build to build; results do not)
      ids[sl] = id;
      if (id < SEGGER_BLOCK_CAP) blk_src[blk * SEGGER_BLOCK_CAP + id] = keys[sl];
    }
  }
  __syncthreads();
  const int n = counter;
  if (tid == 0) {
    blk_cnt[blk] = n <= SEGGER_BLOCK_CAP ? n : -1;
    if (n > SEGGER_BLOCK_CAP) *overflow = 1;
  }
  for (int e = l; e < rlen[g]; e += 16) {
    const int32_t v = col[rbeg[g] + e];
    int h = slot_of(v);
    while (keys[h] != v) h = (h + 1) & (kTabSlots - 1);
    const int id = ids[h];
    col_local[rbeg[g] + e] = (uint8_t)(id < SEGGER_BLOCK_CAP ? id : 0);
  }
}
} }  // namespace segger::(anonymous)

extern "C" int segger_csr_block_tables(const int64_t* indptr, const int32_t* col, const int32_t* row_order, int64_t n_rows,
                                       int64_t n_edges, int32_t* blk_cnt, int32_t* blk_src, uint8_t* col_local, int32_t* overflow,
                                       segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows >= 0 && n_edges >= 0 && n_rows < 0x7fffffffLL, "segger_csr_block_tables: bad sizes");
  if (n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(indptr && blk_cnt && blk_src && overflow && (n_edges == 0 || (col && col_local)), "segger_csr_block_tables: NULL pointer");
  const int64_t nblk = (n_rows + SEGGER_BLOCK_ROWS - 1) / SEGGER_BLOCK_ROWS;
  hipLaunchKernelGGL(csr_block_tables_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, indptr, col, row_order, n_rows,
                     blk_cnt, blk_src, col_local, overflow);
  SEGGER_LAUNCH_CHECK("csr_block_tables_kernel");
  return SEGGER_OK;
}

extern "C" int segger_csr_row_order(const int64_t* indptr, int64_t n_rows, int32_t window, int32_t* order_out,
                                    segger_stream_t stream) {
  SEGGER_REQUIRE(n_rows >= 0 && n_rows < 0x7fffffffLL, "segger_csr_row_order: bad n_rows");
  SEGGER_REQUIRE(window >= 1 && window <= 64 && (window & (window - 1)) == 0,
                 "segger_csr_row_order: window must be a power of two <= 64");
  if (n_rows == 0) return SEGGER_OK;
  SEGGER_REQUIRE(indptr && order_out, "segger_csr_row_order: NULL pointer");
  const int64_t waves = (n_rows + 63) / 64;
  hipLaunchKernelGGL(csr_row_order_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     indptr, n_rows, (int)window, order_out);
  SEGGER_LAUNCH_CHECK("csr_row_order_kernel");
  return SEGGER_OK;
}

extern "C" int segger_coo_unique(const int64_t* ids, int64_t n, int64_t n_ids, int32_t* marks, int32_t* unique_out,
                                 segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n >= 0 && n_ids >= 0, "segger_coo_unique: negative size");
  SEGGER_REQUIRE(unique_out != nullptr, "segger_coo_unique: unique_out is NULL");
  const int32_t one = 1;
  SEGGER_HIP(hipMemsetD32Async((hipDeviceptr_t)unique_out, one, 1, stream));
  if (n == 0) return SEGGER_OK;
  SEGGER_REQUIRE(ids && marks, "segger_coo_unique: NULL pointer");
  SEGGER_HIP(hipMemsetAsync(marks, 0, (size_t)n_ids * sizeof(int32_t), stream));
  hipLaunchKernelGGL(coo_unique_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, n, n_ids, marks,
                     unique_out);
  SEGGER_LAUNCH_CHECK("coo_unique_kernel");
  return SEGGER_OK;
}
