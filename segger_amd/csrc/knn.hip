// Exact k-nearest-neighbour search on a uniform grid (2-D points), the device counterpart of the
// scipy KDTree query segger uses to build tx-neighbors-tx / tx-neighbors-bd edges
// (reference src/segger/data/utils/neighbors.py:122-163: KDTree(points).query(q, k, distance_upper_bound)).
//
//   1. bin every point into a cell of side `cell` (ids clamped to the nx x ny grid), radix-sort point
//      ids by cell (rocPRIM), build cell_start[] and a cell-ordered copy of the coordinates;
//   2. one thread per query walks square rings of cells around its own cell, keeping the k best
//      candidates in a sorted register array.  After ring r every point within r*cell of the query has
//      been seen (floor(a + r) - floor(a) = r), so the walk stops when the k-th distance <= r*cell,
//      when r*cell >= max_dist, or when the ring has left the grid on all sides.
// Output rows are sorted by distance; missing neighbours (fewer than k within max_dist) are padded with
// index n_points and distance +inf, exactly like scipy.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace segger {
namespace {

struct Grid { float x0, y0, inv_cell, cell; int nx, ny; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void knn_bin_kernel(const float* __restrict__ pts, int64_t n, Grid g,
                                                     uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int cx = clampi((int)floorf((pts[2 * i] - g.x0) * g.inv_cell), 0, g.nx - 1);
  const int cy = clampi((int)floorf((pts[2 * i + 1] - g.y0) * g.inv_cell), 0, g.ny - 1);
  keys[i] = (uint32_t)(cy * g.nx + cx);
  vals[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void knn_cells_kernel(const uint32_t* __restrict__ keys_sorted, const int32_t* __restrict__ perm,
                                                       const float* __restrict__ pts, int64_t n, int64_t n_cells,
                                                       int32_t* __restrict__ cell_start, float2* __restrict__ sorted_pts) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const int64_t p = perm[s];
  sorted_pts[s] = make_float2(pts[2 * p], pts[2 * p + 1]);
  const int64_t k = keys_sorted[s];
  const int64_t kprev = s > 0 ? (int64_t)keys_sorted[s - 1] : -1;
  for (int64_t c = kprev + 1; c <= k; ++c) cell_start[c] = (int32_t)s;
  if (s == n - 1)
    for (int64_t c = k + 1; c <= n_cells; ++c) cell_start[c] = (int32_t)n;
}

// K = compile-time capacity (k <= K); bd/bi stay in registers because every index is static
template <int K>
__global__ __launch_bounds__(128) void knn_query_kernel(const float* __restrict__ queries, int64_t m, Grid g, int k, float max_d2,
                                                       const int32_t* __restrict__ cell_start, const float2* __restrict__ sorted_pts,
                                                       const int32_t* __restrict__ perm, int32_t n_points,
                                                       int32_t* __restrict__ nbr, float* __restrict__ dist) {
  const int64_t q = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (q >= m) return;
  const float qx = queries[2 * q], qy = queries[2 * q + 1];
  const int cx = clampi((int)floorf((qx - g.x0) * g.inv_cell), 0, g.nx - 1);
  const int cy = clampi((int)floorf((qy - g.y0) * g.inv_cell), 0, g.ny - 1);
  float bd[K];
  int bi[K];
#pragma unroll
  for (int i = 0; i < K; ++i) { bd[i] = INFINITY; bi[i] = n_points; }
  float kth = INFINITY;                              // k-th best squared distance so far (inf until k found)

  auto visit = [&](int ccx, int ccy) {
    const int c = ccy * g.nx + ccx;
    const int s0 = cell_start[c], s1 = cell_start[c + 1];
    for (int s = s0; s < s1; ++s) {
      const float2 p = sorted_pts[s];
      const float dx = p.x - qx, dy = p.y - qy;
      float cd = dx * dx + dy * dy;
      if (cd <= max_d2 && cd < kth) {
        int ci = perm[s];
#pragma unroll
        for (int i = 0; i < K; ++i) {               // sorted insertion: bubble the displaced entry down
          if (i < k && cd < bd[i]) { const float td = bd[i]; const int ti = bi[i]; bd[i] = cd; bi[i] = ci; cd = td; ci = ti; }
        }
        kth = INFINITY;
#pragma unroll
        for (int i = 0; i < K; ++i) if (i == k - 1) kth = bd[i];
      }
    }
  };

  const int rmax = max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy));
  for (int r = 0; r <= rmax; ++r) {
    if (r == 0) {
      visit(cx, cy);
    } else {
      const int x_lo = cx - r, x_hi = cx + r, y_lo = cy - r, y_hi = cy + r;
      for (int x = max(x_lo, 0); x <= min(x_hi, g.nx - 1); ++x) {
        if (y_lo >= 0) visit(x, y_lo);
        if (y_hi < g.ny) visit(x, y_hi);
      }
      for (int y = max(y_lo + 1, 0); y <= min(y_hi - 1, g.ny - 1); ++y) {
        if (x_lo >= 0) visit(x_lo, y);
        if (x_hi < g.nx) visit(x_hi, y);
      }
    }
    const float reach = (float)r * g.cell;           // every point within `reach` of the query has been seen
    const float reach2 = reach * reach;
    if (reach2 >= max_d2 || kth <= reach2) break;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    if (i < k) {
      nbr[q * k + i] = bi[i];
      if (dist) dist[q * k + i] = bi[i] == n_points ? INFINITY : sqrtf(bd[i]);
    }
  }
}

int key_bits(int64_t n_cells) {
  int b = 1;
  while (b < 32 && (1LL << b) < n_cells) ++b;
  return b;
}
size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
size_t sort_temp_bytes(int64_t n, int64_t n_cells) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, key_bits(n_cells), (hipStream_t)0);
  return bytes;
}

}  // namespace
}  // namespace segger

using namespace segger;

extern "C" size_t segger_knn_workspace_bytes(int64_t n_points, int32_t nx, int32_t ny) {
  if (n_points <= 0 || nx <= 0 || ny <= 0) return 256;
  const int64_t n_cells = (int64_t)nx * ny;
  // keys_in, keys_out, vals_in, perm (4 B each) + sorted points (8 B) + cell_start + sort temp
  return 4 * align_up((size_t)n_points * 4) + align_up((size_t)n_points * 8) + align_up((size_t)(n_cells + 1) * 4) +
         align_up(sort_temp_bytes(n_points, n_cells)) + 256;
}

extern "C" int segger_knn_grid(const float* points, int64_t n_points, const float* queries, int64_t n_queries, int32_t k,
                               float max_dist, float x0, float y0, float cell, int32_t nx, int32_t ny,
                               int32_t* nbr, float* dist, void* workspace, size_t workspace_bytes, segger_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGGER_REQUIRE(n_points >= 0 && n_queries >= 0, "segger_knn_grid: negative size");
  SEGGER_REQUIRE(k >= 1 && k <= 64, "segger_knn_grid: k must be in [1, 64]");
  SEGGER_REQUIRE(n_points < 0x7fffffffLL && n_queries * (int64_t)k < (1LL << 40), "segger_knn_grid: too many points");
  if (n_queries == 0) return SEGGER_OK;
  SEGGER_REQUIRE(nbr != nullptr, "segger_knn_grid: nbr is NULL");
  if (queries == nullptr) { queries = points; SEGGER_REQUIRE(n_queries == n_points, "segger_knn_grid: queries NULL but n_queries != n_points"); }
  SEGGER_REQUIRE(n_points == 0 || points != nullptr, "segger_knn_grid: points is NULL");
  SEGGER_REQUIRE(cell > 0.f && nx > 0 && ny > 0 && (int64_t)nx * ny < 0x7fffffffLL, "segger_knn_grid: bad grid");
  SEGGER_REQUIRE(max_dist > 0.f, "segger_knn_grid: max_dist must be positive (use +inf for none)");
  const int64_t n_cells = (int64_t)nx * ny;
  const size_t need = segger_knn_workspace_bytes(n_points, nx, ny);
  if (!workspace || workspace_bytes < need) {
    set_error("segger_knn_grid: workspace %zu < %zu bytes", workspace_bytes, need);
    return SEGGER_EWORKSPACE;
  }
  char* base = static_cast<char*>(workspace);
  const size_t seg = align_up((size_t)(n_points > 0 ? n_points : 1) * 4);
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(base);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + seg);
  int32_t* vals_in = reinterpret_cast<int32_t*>(base + 2 * seg);
  int32_t* perm = reinterpret_cast<int32_t*>(base + 3 * seg);
  float2* sorted_pts = reinterpret_cast<float2*>(base + 4 * seg);
  int32_t* cell_start = reinterpret_cast<int32_t*>(base + 4 * seg + align_up((size_t)(n_points > 0 ? n_points : 1) * 8));
  void* temp = reinterpret_cast<char*>(cell_start) + align_up((size_t)(n_cells + 1) * 4);
  Grid g{x0, y0, 1.0f / cell, cell, nx, ny};
  if (n_points > 0) {
    const unsigned nb = (unsigned)((n_points + 255) / 256);
    hipLaunchKernelGGL(knn_bin_kernel, dim3(nb), dim3(256), 0, stream, points, n_points, g, keys_in, vals_in);
    size_t temp_bytes = sort_temp_bytes(n_points, n_cells);
    SEGGER_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, perm, (size_t)n_points, 0,
                                         key_bits(n_cells), stream));
    hipLaunchKernelGGL(knn_cells_kernel, dim3(nb), dim3(256), 0, stream, keys_out, perm, points, n_points, n_cells,
                       cell_start, sorted_pts);
  } else {
    SEGGER_HIP(hipMemsetAsync(cell_start, 0, (size_t)(n_cells + 1) * 4, stream));
  }
  const float max_d2 = max_dist * max_dist;          // inf stays inf
  const unsigned nq = (unsigned)((n_queries + 127) / 128);
#define GO(K) hipLaunchKernelGGL((knn_query_kernel<K>), dim3(nq), dim3(128), 0, stream, queries, n_queries, g, (int)k, max_d2, \
                                 cell_start, sorted_pts, perm, (int32_t)n_points, nbr, dist)
  if (k <= 8) GO(8); else if (k <= 16) GO(16); else if (k <= 32) GO(32); else GO(64);
#undef GO
  SEGGER_LAUNCH_CHECK("knn kernels");
  return SEGGER_OK;
}
