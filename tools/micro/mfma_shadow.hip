// How many independent VALU instructions does ONE wave per SIMD issue in the shadow of its own MFMAs?
// hipcc --offload-arch=gfx950 -O3 mfma_shadow.hip -o mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int CHAINS>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* a, float* y, int iters) {
  __shared__ float pad[36 * 1024];                       // 144 KB: one workgroup per CU
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  const bf16x8 av = a[threadIdx.x], bv = a[threadIdx.x + 256];
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x + i;
  pad[threadIdx.x] = 0.f;
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      acc[m % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m % CHAINS], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i % 16]));
    }
  }
  const long long c1 = clock64();
  if (blockIdx.x == 3 && threadIdx.x == 0) printf("   clock64 per MFMA slot: %.1f\n", (double)(c1 - c0) / (iters * 6.0));
  float s = pad[threadIdx.x];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  y[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int CHAINS>
void run(const bf16x8* a, float* y) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, CHAINS>), dim3(256), dim3(256), 0, 0, a, y, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  const double per_mfma_ns = ms * 1e6 / (iters * 6.0);
  printf("chains %d  VALU per MFMA %2d: %.3f ms, %.2f ns per MFMA slot (32 cycles = %.2f ns at 2.4 GHz)\n", CHAINS, NV, ms, per_mfma_ns, 32 / 2.4);
}
int main() {
  bf16x8* a; float* y;
  hipMalloc(&a, 512 * 16); hipMemset(a, 0, 512 * 16); hipMalloc(&y, 256 * 256 * 4);
  printf("-- zero operands\n");
  run<0, 1>(a, y); run<4, 1>(a, y);
  {                                                    // random bf16 bit patterns of moderate magnitude: the matrix pipe's power draw is data dependent
    unsigned short h[512 * 8];
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 16) & 0x83ffu)); }
    hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
  }
  printf("-- random operands\n");
  run<0, 1>(a, y); run<2, 1>(a, y); run<4, 1>(a, y); run<6, 1>(a, y); run<8, 1>(a, y); run<12, 1>(a, y); run<16, 1>(a, y);
  run<0, 2>(a, y); run<4, 2>(a, y); run<6, 2>(a, y); run<8, 2>(a, y);
  return 0;
}
