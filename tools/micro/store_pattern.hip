// Output-store patterns of a [n, 128] fp32 matrix, one wave per 32 rows x 32 columns (the MFMA 32x32 C layout of the fp32
// projection kernels) against row-contiguous stores.  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// (a) MFMA C layout: lane (j = lane & 31, h = lane >> 5) stores row j, columns 8 g + 4 h .. + 3 for g = 0..3: 32 bytes per row
//     and instruction
__global__ __launch_bounds__(256) void mfma_layout(float* y, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    float* yr = y + ((size_t)t * 32 + j) * 128 + wave * 32 + 4 * h;
    const f32x4 v = {(float)t, (float)j, 1.f, 2.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(yr + 8 * g) = v;
  }
}
// (b) row-contiguous: 8 lanes x 16 bytes = the wave's 128 bytes of a row per instruction, 8 rows per instruction
__global__ __launch_bounds__(256) void row_contiguous(float* y, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    float* yr = y + ((size_t)t * 32 + (lane >> 3)) * 128 + wave * 32 + 4 * (lane & 7);
    const f32x4 v = {(float)t, (float)lane, 1.f, 2.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(yr + (size_t)8 * g * 128) = v;
  }
}
// (c) a whole 512-byte row per 32 lanes (the 4 waves of (b) merged: what one wave owning all 128 columns could do)
__global__ __launch_bounds__(256) void full_rows(float* y, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    float* yr = y + ((size_t)t * 32 + wave * 8 + (lane >> 5)) * 128 + 4 * (lane & 31);
    const f32x4 v = {(float)t, (float)lane, 1.f, 2.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(yr + (size_t)2 * g * 128) = v;
  }
}
int main() {
  const int n = 1000000, n_tiles = n / 32;
  float* y; hipMalloc(&y, (size_t)n * 128 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {256, 512, 2048, n_tiles}) {
    for (int k = 0; k < 3; ++k) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it) {
          if (k == 0) hipLaunchKernelGGL(mfma_layout, dim3(grid), dim3(256), 0, 0, y, n_tiles);
          if (k == 1) hipLaunchKernelGGL(row_contiguous, dim3(grid), dim3(256), 0, 0, y, n_tiles);
          if (k == 2) hipLaunchKernelGGL(full_rows, dim3(grid), dim3(256), 0, 0, y, n_tiles);
        }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      }
      printf("grid %6d %-15s %.3f ms  %.2f TB/s\n", grid, k == 0 ? "mfma_layout" : k == 1 ? "row_contiguous" : "full_rows", ms / 10, 0.512 / (ms / 10) );
    }
  }
  return 0;
}
