// Issue-cost probe for the VALU instructions the GATv2 kernels are made of (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o build/valu_probe && build/valu_probe
// Every kernel issues REPS x 32 copies of ONE instruction per wave, cycling over 8 independent destination
// registers; the table is SIMD cycles per wave-instruction at 1, 2, 4 and 8 resident waves per SIMD
// (2.4 GHz assumed: the ratios between rows are what matters).  The kernels in csrc/gatv2_kernels.h are
// VALU-bound, so these prices decide which formulation of an inner loop is cheapest.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REPS 2048

__global__ __launch_bounds__(256) void probe_0(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_fma_f32 %0, %[b], %[c], %0\n"
                 "v_fma_f32 %1, %[b], %[c], %1\n"
                 "v_fma_f32 %2, %[b], %[c], %2\n"
                 "v_fma_f32 %3, %[b], %[c], %3\n"
                 "v_fma_f32 %4, %[b], %[c], %4\n"
                 "v_fma_f32 %5, %[b], %[c], %5\n"
                 "v_fma_f32 %6, %[b], %[c], %6\n"
                 "v_fma_f32 %7, %[b], %[c], %7\n"
                 "v_fma_f32 %0, %[b], %[c], %0\n"
                 "v_fma_f32 %1, %[b], %[c], %1\n"
                 "v_fma_f32 %2, %[b], %[c], %2\n"
                 "v_fma_f32 %3, %[b], %[c], %3\n"
                 "v_fma_f32 %4, %[b], %[c], %4\n"
                 "v_fma_f32 %5, %[b], %[c], %5\n"
                 "v_fma_f32 %6, %[b], %[c], %6\n"
                 "v_fma_f32 %7, %[b], %[c], %7\n"
                 "v_fma_f32 %0, %[b], %[c], %0\n"
                 "v_fma_f32 %1, %[b], %[c], %1\n"
                 "v_fma_f32 %2, %[b], %[c], %2\n"
                 "v_fma_f32 %3, %[b], %[c], %3\n"
                 "v_fma_f32 %4, %[b], %[c], %4\n"
                 "v_fma_f32 %5, %[b], %[c], %5\n"
                 "v_fma_f32 %6, %[b], %[c], %6\n"
                 "v_fma_f32 %7, %[b], %[c], %7\n"
                 "v_fma_f32 %0, %[b], %[c], %0\n"
                 "v_fma_f32 %1, %[b], %[c], %1\n"
                 "v_fma_f32 %2, %[b], %[c], %2\n"
                 "v_fma_f32 %3, %[b], %[c], %3\n"
                 "v_fma_f32 %4, %[b], %[c], %4\n"
                 "v_fma_f32 %5, %[b], %[c], %5\n"
                 "v_fma_f32 %6, %[b], %[c], %6\n"
                 "v_fma_f32 %7, %[b], %[c], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_1(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_fma_f32 %0, %[b], |%[c]|, %0\n"
                 "v_fma_f32 %1, %[b], |%[c]|, %1\n"
                 "v_fma_f32 %2, %[b], |%[c]|, %2\n"
                 "v_fma_f32 %3, %[b], |%[c]|, %3\n"
                 "v_fma_f32 %4, %[b], |%[c]|, %4\n"
                 "v_fma_f32 %5, %[b], |%[c]|, %5\n"
                 "v_fma_f32 %6, %[b], |%[c]|, %6\n"
                 "v_fma_f32 %7, %[b], |%[c]|, %7\n"
                 "v_fma_f32 %0, %[b], |%[c]|, %0\n"
                 "v_fma_f32 %1, %[b], |%[c]|, %1\n"
                 "v_fma_f32 %2, %[b], |%[c]|, %2\n"
                 "v_fma_f32 %3, %[b], |%[c]|, %3\n"
                 "v_fma_f32 %4, %[b], |%[c]|, %4\n"
                 "v_fma_f32 %5, %[b], |%[c]|, %5\n"
                 "v_fma_f32 %6, %[b], |%[c]|, %6\n"
                 "v_fma_f32 %7, %[b], |%[c]|, %7\n"
                 "v_fma_f32 %0, %[b], |%[c]|, %0\n"
                 "v_fma_f32 %1, %[b], |%[c]|, %1\n"
                 "v_fma_f32 %2, %[b], |%[c]|, %2\n"
                 "v_fma_f32 %3, %[b], |%[c]|, %3\n"
                 "v_fma_f32 %4, %[b], |%[c]|, %4\n"
                 "v_fma_f32 %5, %[b], |%[c]|, %5\n"
                 "v_fma_f32 %6, %[b], |%[c]|, %6\n"
                 "v_fma_f32 %7, %[b], |%[c]|, %7\n"
                 "v_fma_f32 %0, %[b], |%[c]|, %0\n"
                 "v_fma_f32 %1, %[b], |%[c]|, %1\n"
                 "v_fma_f32 %2, %[b], |%[c]|, %2\n"
                 "v_fma_f32 %3, %[b], |%[c]|, %3\n"
                 "v_fma_f32 %4, %[b], |%[c]|, %4\n"
                 "v_fma_f32 %5, %[b], |%[c]|, %5\n"
                 "v_fma_f32 %6, %[b], |%[c]|, %6\n"
                 "v_fma_f32 %7, %[b], |%[c]|, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_2(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_add_f32 %0, %[b], %0\n"
                 "v_add_f32 %1, %[b], %1\n"
                 "v_add_f32 %2, %[b], %2\n"
                 "v_add_f32 %3, %[b], %3\n"
                 "v_add_f32 %4, %[b], %4\n"
                 "v_add_f32 %5, %[b], %5\n"
                 "v_add_f32 %6, %[b], %6\n"
                 "v_add_f32 %7, %[b], %7\n"
                 "v_add_f32 %0, %[b], %0\n"
                 "v_add_f32 %1, %[b], %1\n"
                 "v_add_f32 %2, %[b], %2\n"
                 "v_add_f32 %3, %[b], %3\n"
                 "v_add_f32 %4, %[b], %4\n"
                 "v_add_f32 %5, %[b], %5\n"
                 "v_add_f32 %6, %[b], %6\n"
                 "v_add_f32 %7, %[b], %7\n"
                 "v_add_f32 %0, %[b], %0\n"
                 "v_add_f32 %1, %[b], %1\n"
                 "v_add_f32 %2, %[b], %2\n"
                 "v_add_f32 %3, %[b], %3\n"
                 "v_add_f32 %4, %[b], %4\n"
                 "v_add_f32 %5, %[b], %5\n"
                 "v_add_f32 %6, %[b], %6\n"
                 "v_add_f32 %7, %[b], %7\n"
                 "v_add_f32 %0, %[b], %0\n"
                 "v_add_f32 %1, %[b], %1\n"
                 "v_add_f32 %2, %[b], %2\n"
                 "v_add_f32 %3, %[b], %3\n"
                 "v_add_f32 %4, %[b], %4\n"
                 "v_add_f32 %5, %[b], %5\n"
                 "v_add_f32 %6, %[b], %6\n"
                 "v_add_f32 %7, %[b], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_3(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mul_f32 %0, %[b], %0\n"
                 "v_mul_f32 %1, %[b], %1\n"
                 "v_mul_f32 %2, %[b], %2\n"
                 "v_mul_f32 %3, %[b], %3\n"
                 "v_mul_f32 %4, %[b], %4\n"
                 "v_mul_f32 %5, %[b], %5\n"
                 "v_mul_f32 %6, %[b], %6\n"
                 "v_mul_f32 %7, %[b], %7\n"
                 "v_mul_f32 %0, %[b], %0\n"
                 "v_mul_f32 %1, %[b], %1\n"
                 "v_mul_f32 %2, %[b], %2\n"
                 "v_mul_f32 %3, %[b], %3\n"
                 "v_mul_f32 %4, %[b], %4\n"
                 "v_mul_f32 %5, %[b], %5\n"
                 "v_mul_f32 %6, %[b], %6\n"
                 "v_mul_f32 %7, %[b], %7\n"
                 "v_mul_f32 %0, %[b], %0\n"
                 "v_mul_f32 %1, %[b], %1\n"
                 "v_mul_f32 %2, %[b], %2\n"
                 "v_mul_f32 %3, %[b], %3\n"
                 "v_mul_f32 %4, %[b], %4\n"
                 "v_mul_f32 %5, %[b], %5\n"
                 "v_mul_f32 %6, %[b], %6\n"
                 "v_mul_f32 %7, %[b], %7\n"
                 "v_mul_f32 %0, %[b], %0\n"
                 "v_mul_f32 %1, %[b], %1\n"
                 "v_mul_f32 %2, %[b], %2\n"
                 "v_mul_f32 %3, %[b], %3\n"
                 "v_mul_f32 %4, %[b], %4\n"
                 "v_mul_f32 %5, %[b], %5\n"
                 "v_mul_f32 %6, %[b], %6\n"
                 "v_mul_f32 %7, %[b], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_4(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_max_f32 %0, %[b], %0\n"
                 "v_max_f32 %1, %[b], %1\n"
                 "v_max_f32 %2, %[b], %2\n"
                 "v_max_f32 %3, %[b], %3\n"
                 "v_max_f32 %4, %[b], %4\n"
                 "v_max_f32 %5, %[b], %5\n"
                 "v_max_f32 %6, %[b], %6\n"
                 "v_max_f32 %7, %[b], %7\n"
                 "v_max_f32 %0, %[b], %0\n"
                 "v_max_f32 %1, %[b], %1\n"
                 "v_max_f32 %2, %[b], %2\n"
                 "v_max_f32 %3, %[b], %3\n"
                 "v_max_f32 %4, %[b], %4\n"
                 "v_max_f32 %5, %[b], %5\n"
                 "v_max_f32 %6, %[b], %6\n"
                 "v_max_f32 %7, %[b], %7\n"
                 "v_max_f32 %0, %[b], %0\n"
                 "v_max_f32 %1, %[b], %1\n"
                 "v_max_f32 %2, %[b], %2\n"
                 "v_max_f32 %3, %[b], %3\n"
                 "v_max_f32 %4, %[b], %4\n"
                 "v_max_f32 %5, %[b], %5\n"
                 "v_max_f32 %6, %[b], %6\n"
                 "v_max_f32 %7, %[b], %7\n"
                 "v_max_f32 %0, %[b], %0\n"
                 "v_max_f32 %1, %[b], %1\n"
                 "v_max_f32 %2, %[b], %2\n"
                 "v_max_f32 %3, %[b], %3\n"
                 "v_max_f32 %4, %[b], %4\n"
                 "v_max_f32 %5, %[b], %5\n"
                 "v_max_f32 %6, %[b], %6\n"
                 "v_max_f32 %7, %[b], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_5(float* out) {
  float2 a0 = {threadIdx.x * 1e-3f, 0.f}; float2 a1 = {threadIdx.x * 1e-3f, 1.f}; float2 a2 = {threadIdx.x * 1e-3f, 2.f}; float2 a3 = {threadIdx.x * 1e-3f, 3.f}; float2 a4 = {threadIdx.x * 1e-3f, 4.f}; float2 a5 = {threadIdx.x * 1e-3f, 5.f}; float2 a6 = {threadIdx.x * 1e-3f, 6.f}; float2 a7 = {threadIdx.x * 1e-3f, 7.f};
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_pk_fma_f32 %0, %[q], %[r], %0\n"
                 "v_pk_fma_f32 %1, %[q], %[r], %1\n"
                 "v_pk_fma_f32 %2, %[q], %[r], %2\n"
                 "v_pk_fma_f32 %3, %[q], %[r], %3\n"
                 "v_pk_fma_f32 %4, %[q], %[r], %4\n"
                 "v_pk_fma_f32 %5, %[q], %[r], %5\n"
                 "v_pk_fma_f32 %6, %[q], %[r], %6\n"
                 "v_pk_fma_f32 %7, %[q], %[r], %7\n"
                 "v_pk_fma_f32 %0, %[q], %[r], %0\n"
                 "v_pk_fma_f32 %1, %[q], %[r], %1\n"
                 "v_pk_fma_f32 %2, %[q], %[r], %2\n"
                 "v_pk_fma_f32 %3, %[q], %[r], %3\n"
                 "v_pk_fma_f32 %4, %[q], %[r], %4\n"
                 "v_pk_fma_f32 %5, %[q], %[r], %5\n"
                 "v_pk_fma_f32 %6, %[q], %[r], %6\n"
                 "v_pk_fma_f32 %7, %[q], %[r], %7\n"
                 "v_pk_fma_f32 %0, %[q], %[r], %0\n"
                 "v_pk_fma_f32 %1, %[q], %[r], %1\n"
                 "v_pk_fma_f32 %2, %[q], %[r], %2\n"
                 "v_pk_fma_f32 %3, %[q], %[r], %3\n"
                 "v_pk_fma_f32 %4, %[q], %[r], %4\n"
                 "v_pk_fma_f32 %5, %[q], %[r], %5\n"
                 "v_pk_fma_f32 %6, %[q], %[r], %6\n"
                 "v_pk_fma_f32 %7, %[q], %[r], %7\n"
                 "v_pk_fma_f32 %0, %[q], %[r], %0\n"
                 "v_pk_fma_f32 %1, %[q], %[r], %1\n"
                 "v_pk_fma_f32 %2, %[q], %[r], %2\n"
                 "v_pk_fma_f32 %3, %[q], %[r], %3\n"
                 "v_pk_fma_f32 %4, %[q], %[r], %4\n"
                 "v_pk_fma_f32 %5, %[q], %[r], %5\n"
                 "v_pk_fma_f32 %6, %[q], %[r], %6\n"
                 "v_pk_fma_f32 %7, %[q], %[r], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0.x == 12345.f && a1.x == 12345.f && a2.x == 12345.f && a3.x == 12345.f && a4.x == 12345.f && a5.x == 12345.f && a6.x == 12345.f && a7.x == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_6(float* out) {
  float2 a0 = {threadIdx.x * 1e-3f, 0.f}; float2 a1 = {threadIdx.x * 1e-3f, 1.f}; float2 a2 = {threadIdx.x * 1e-3f, 2.f}; float2 a3 = {threadIdx.x * 1e-3f, 3.f}; float2 a4 = {threadIdx.x * 1e-3f, 4.f}; float2 a5 = {threadIdx.x * 1e-3f, 5.f}; float2 a6 = {threadIdx.x * 1e-3f, 6.f}; float2 a7 = {threadIdx.x * 1e-3f, 7.f};
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_pk_add_f32 %0, %[q], %0\n"
                 "v_pk_add_f32 %1, %[q], %1\n"
                 "v_pk_add_f32 %2, %[q], %2\n"
                 "v_pk_add_f32 %3, %[q], %3\n"
                 "v_pk_add_f32 %4, %[q], %4\n"
                 "v_pk_add_f32 %5, %[q], %5\n"
                 "v_pk_add_f32 %6, %[q], %6\n"
                 "v_pk_add_f32 %7, %[q], %7\n"
                 "v_pk_add_f32 %0, %[q], %0\n"
                 "v_pk_add_f32 %1, %[q], %1\n"
                 "v_pk_add_f32 %2, %[q], %2\n"
                 "v_pk_add_f32 %3, %[q], %3\n"
                 "v_pk_add_f32 %4, %[q], %4\n"
                 "v_pk_add_f32 %5, %[q], %5\n"
                 "v_pk_add_f32 %6, %[q], %6\n"
                 "v_pk_add_f32 %7, %[q], %7\n"
                 "v_pk_add_f32 %0, %[q], %0\n"
                 "v_pk_add_f32 %1, %[q], %1\n"
                 "v_pk_add_f32 %2, %[q], %2\n"
                 "v_pk_add_f32 %3, %[q], %3\n"
                 "v_pk_add_f32 %4, %[q], %4\n"
                 "v_pk_add_f32 %5, %[q], %5\n"
                 "v_pk_add_f32 %6, %[q], %6\n"
                 "v_pk_add_f32 %7, %[q], %7\n"
                 "v_pk_add_f32 %0, %[q], %0\n"
                 "v_pk_add_f32 %1, %[q], %1\n"
                 "v_pk_add_f32 %2, %[q], %2\n"
                 "v_pk_add_f32 %3, %[q], %3\n"
                 "v_pk_add_f32 %4, %[q], %4\n"
                 "v_pk_add_f32 %5, %[q], %5\n"
                 "v_pk_add_f32 %6, %[q], %6\n"
                 "v_pk_add_f32 %7, %[q], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0.x == 12345.f && a1.x == 12345.f && a2.x == 12345.f && a3.x == 12345.f && a4.x == 12345.f && a5.x == 12345.f && a6.x == 12345.f && a7.x == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_7(float* out) {
  float2 a0 = {threadIdx.x * 1e-3f, 0.f}; float2 a1 = {threadIdx.x * 1e-3f, 1.f}; float2 a2 = {threadIdx.x * 1e-3f, 2.f}; float2 a3 = {threadIdx.x * 1e-3f, 3.f}; float2 a4 = {threadIdx.x * 1e-3f, 4.f}; float2 a5 = {threadIdx.x * 1e-3f, 5.f}; float2 a6 = {threadIdx.x * 1e-3f, 6.f}; float2 a7 = {threadIdx.x * 1e-3f, 7.f};
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_pk_mul_f32 %0, %[q], %0\n"
                 "v_pk_mul_f32 %1, %[q], %1\n"
                 "v_pk_mul_f32 %2, %[q], %2\n"
                 "v_pk_mul_f32 %3, %[q], %3\n"
                 "v_pk_mul_f32 %4, %[q], %4\n"
                 "v_pk_mul_f32 %5, %[q], %5\n"
                 "v_pk_mul_f32 %6, %[q], %6\n"
                 "v_pk_mul_f32 %7, %[q], %7\n"
                 "v_pk_mul_f32 %0, %[q], %0\n"
                 "v_pk_mul_f32 %1, %[q], %1\n"
                 "v_pk_mul_f32 %2, %[q], %2\n"
                 "v_pk_mul_f32 %3, %[q], %3\n"
                 "v_pk_mul_f32 %4, %[q], %4\n"
                 "v_pk_mul_f32 %5, %[q], %5\n"
                 "v_pk_mul_f32 %6, %[q], %6\n"
                 "v_pk_mul_f32 %7, %[q], %7\n"
                 "v_pk_mul_f32 %0, %[q], %0\n"
                 "v_pk_mul_f32 %1, %[q], %1\n"
                 "v_pk_mul_f32 %2, %[q], %2\n"
                 "v_pk_mul_f32 %3, %[q], %3\n"
                 "v_pk_mul_f32 %4, %[q], %4\n"
                 "v_pk_mul_f32 %5, %[q], %5\n"
                 "v_pk_mul_f32 %6, %[q], %6\n"
                 "v_pk_mul_f32 %7, %[q], %7\n"
                 "v_pk_mul_f32 %0, %[q], %0\n"
                 "v_pk_mul_f32 %1, %[q], %1\n"
                 "v_pk_mul_f32 %2, %[q], %2\n"
                 "v_pk_mul_f32 %3, %[q], %3\n"
                 "v_pk_mul_f32 %4, %[q], %4\n"
                 "v_pk_mul_f32 %5, %[q], %5\n"
                 "v_pk_mul_f32 %6, %[q], %6\n"
                 "v_pk_mul_f32 %7, %[q], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0.x == 12345.f && a1.x == 12345.f && a2.x == 12345.f && a3.x == 12345.f && a4.x == 12345.f && a5.x == 12345.f && a6.x == 12345.f && a7.x == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_8(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_and_b32 %0, %[v], %0\n"
                 "v_and_b32 %1, %[v], %1\n"
                 "v_and_b32 %2, %[v], %2\n"
                 "v_and_b32 %3, %[v], %3\n"
                 "v_and_b32 %4, %[v], %4\n"
                 "v_and_b32 %5, %[v], %5\n"
                 "v_and_b32 %6, %[v], %6\n"
                 "v_and_b32 %7, %[v], %7\n"
                 "v_and_b32 %0, %[v], %0\n"
                 "v_and_b32 %1, %[v], %1\n"
                 "v_and_b32 %2, %[v], %2\n"
                 "v_and_b32 %3, %[v], %3\n"
                 "v_and_b32 %4, %[v], %4\n"
                 "v_and_b32 %5, %[v], %5\n"
                 "v_and_b32 %6, %[v], %6\n"
                 "v_and_b32 %7, %[v], %7\n"
                 "v_and_b32 %0, %[v], %0\n"
                 "v_and_b32 %1, %[v], %1\n"
                 "v_and_b32 %2, %[v], %2\n"
                 "v_and_b32 %3, %[v], %3\n"
                 "v_and_b32 %4, %[v], %4\n"
                 "v_and_b32 %5, %[v], %5\n"
                 "v_and_b32 %6, %[v], %6\n"
                 "v_and_b32 %7, %[v], %7\n"
                 "v_and_b32 %0, %[v], %0\n"
                 "v_and_b32 %1, %[v], %1\n"
                 "v_and_b32 %2, %[v], %2\n"
                 "v_and_b32 %3, %[v], %3\n"
                 "v_and_b32 %4, %[v], %4\n"
                 "v_and_b32 %5, %[v], %5\n"
                 "v_and_b32 %6, %[v], %6\n"
                 "v_and_b32 %7, %[v], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_9(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_lshlrev_b32 %0, 16, %0\n"
                 "v_lshlrev_b32 %1, 16, %1\n"
                 "v_lshlrev_b32 %2, 16, %2\n"
                 "v_lshlrev_b32 %3, 16, %3\n"
                 "v_lshlrev_b32 %4, 16, %4\n"
                 "v_lshlrev_b32 %5, 16, %5\n"
                 "v_lshlrev_b32 %6, 16, %6\n"
                 "v_lshlrev_b32 %7, 16, %7\n"
                 "v_lshlrev_b32 %0, 16, %0\n"
                 "v_lshlrev_b32 %1, 16, %1\n"
                 "v_lshlrev_b32 %2, 16, %2\n"
                 "v_lshlrev_b32 %3, 16, %3\n"
                 "v_lshlrev_b32 %4, 16, %4\n"
                 "v_lshlrev_b32 %5, 16, %5\n"
                 "v_lshlrev_b32 %6, 16, %6\n"
                 "v_lshlrev_b32 %7, 16, %7\n"
                 "v_lshlrev_b32 %0, 16, %0\n"
                 "v_lshlrev_b32 %1, 16, %1\n"
                 "v_lshlrev_b32 %2, 16, %2\n"
                 "v_lshlrev_b32 %3, 16, %3\n"
                 "v_lshlrev_b32 %4, 16, %4\n"
                 "v_lshlrev_b32 %5, 16, %5\n"
                 "v_lshlrev_b32 %6, 16, %6\n"
                 "v_lshlrev_b32 %7, 16, %7\n"
                 "v_lshlrev_b32 %0, 16, %0\n"
                 "v_lshlrev_b32 %1, 16, %1\n"
                 "v_lshlrev_b32 %2, 16, %2\n"
                 "v_lshlrev_b32 %3, 16, %3\n"
                 "v_lshlrev_b32 %4, 16, %4\n"
                 "v_lshlrev_b32 %5, 16, %5\n"
                 "v_lshlrev_b32 %6, 16, %6\n"
                 "v_lshlrev_b32 %7, 16, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_10(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_xor_b32 %0, %[v], %0\n"
                 "v_xor_b32 %1, %[v], %1\n"
                 "v_xor_b32 %2, %[v], %2\n"
                 "v_xor_b32 %3, %[v], %3\n"
                 "v_xor_b32 %4, %[v], %4\n"
                 "v_xor_b32 %5, %[v], %5\n"
                 "v_xor_b32 %6, %[v], %6\n"
                 "v_xor_b32 %7, %[v], %7\n"
                 "v_xor_b32 %0, %[v], %0\n"
                 "v_xor_b32 %1, %[v], %1\n"
                 "v_xor_b32 %2, %[v], %2\n"
                 "v_xor_b32 %3, %[v], %3\n"
                 "v_xor_b32 %4, %[v], %4\n"
                 "v_xor_b32 %5, %[v], %5\n"
                 "v_xor_b32 %6, %[v], %6\n"
                 "v_xor_b32 %7, %[v], %7\n"
                 "v_xor_b32 %0, %[v], %0\n"
                 "v_xor_b32 %1, %[v], %1\n"
                 "v_xor_b32 %2, %[v], %2\n"
                 "v_xor_b32 %3, %[v], %3\n"
                 "v_xor_b32 %4, %[v], %4\n"
                 "v_xor_b32 %5, %[v], %5\n"
                 "v_xor_b32 %6, %[v], %6\n"
                 "v_xor_b32 %7, %[v], %7\n"
                 "v_xor_b32 %0, %[v], %0\n"
                 "v_xor_b32 %1, %[v], %1\n"
                 "v_xor_b32 %2, %[v], %2\n"
                 "v_xor_b32 %3, %[v], %3\n"
                 "v_xor_b32 %4, %[v], %4\n"
                 "v_xor_b32 %5, %[v], %5\n"
                 "v_xor_b32 %6, %[v], %6\n"
                 "v_xor_b32 %7, %[v], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_11(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_add_u32 %0, %[v], %0\n"
                 "v_add_u32 %1, %[v], %1\n"
                 "v_add_u32 %2, %[v], %2\n"
                 "v_add_u32 %3, %[v], %3\n"
                 "v_add_u32 %4, %[v], %4\n"
                 "v_add_u32 %5, %[v], %5\n"
                 "v_add_u32 %6, %[v], %6\n"
                 "v_add_u32 %7, %[v], %7\n"
                 "v_add_u32 %0, %[v], %0\n"
                 "v_add_u32 %1, %[v], %1\n"
                 "v_add_u32 %2, %[v], %2\n"
                 "v_add_u32 %3, %[v], %3\n"
                 "v_add_u32 %4, %[v], %4\n"
                 "v_add_u32 %5, %[v], %5\n"
                 "v_add_u32 %6, %[v], %6\n"
                 "v_add_u32 %7, %[v], %7\n"
                 "v_add_u32 %0, %[v], %0\n"
                 "v_add_u32 %1, %[v], %1\n"
                 "v_add_u32 %2, %[v], %2\n"
                 "v_add_u32 %3, %[v], %3\n"
                 "v_add_u32 %4, %[v], %4\n"
                 "v_add_u32 %5, %[v], %5\n"
                 "v_add_u32 %6, %[v], %6\n"
                 "v_add_u32 %7, %[v], %7\n"
                 "v_add_u32 %0, %[v], %0\n"
                 "v_add_u32 %1, %[v], %1\n"
                 "v_add_u32 %2, %[v], %2\n"
                 "v_add_u32 %3, %[v], %3\n"
                 "v_add_u32 %4, %[v], %4\n"
                 "v_add_u32 %5, %[v], %5\n"
                 "v_add_u32 %6, %[v], %6\n"
                 "v_add_u32 %7, %[v], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_12(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_bitop3_b32 %0, %0, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %1, %1, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %2, %2, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %3, %3, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %4, %4, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %5, %5, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %6, %6, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %7, %7, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %0, %0, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %1, %1, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %2, %2, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %3, %3, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %4, %4, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %5, %5, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %6, %6, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %7, %7, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %0, %0, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %1, %1, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %2, %2, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %3, %3, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %4, %4, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %5, %5, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %6, %6, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %7, %7, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %0, %0, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %1, %1, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %2, %2, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %3, %3, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %4, %4, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %5, %5, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %6, %6, %[v], %[v] bitop3:0x78\n"
                 "v_bitop3_b32 %7, %7, %[v], %[v] bitop3:0x78\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_13(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mul_lo_u32 %0, %0, %[v]\n"
                 "v_mul_lo_u32 %1, %1, %[v]\n"
                 "v_mul_lo_u32 %2, %2, %[v]\n"
                 "v_mul_lo_u32 %3, %3, %[v]\n"
                 "v_mul_lo_u32 %4, %4, %[v]\n"
                 "v_mul_lo_u32 %5, %5, %[v]\n"
                 "v_mul_lo_u32 %6, %6, %[v]\n"
                 "v_mul_lo_u32 %7, %7, %[v]\n"
                 "v_mul_lo_u32 %0, %0, %[v]\n"
                 "v_mul_lo_u32 %1, %1, %[v]\n"
                 "v_mul_lo_u32 %2, %2, %[v]\n"
                 "v_mul_lo_u32 %3, %3, %[v]\n"
                 "v_mul_lo_u32 %4, %4, %[v]\n"
                 "v_mul_lo_u32 %5, %5, %[v]\n"
                 "v_mul_lo_u32 %6, %6, %[v]\n"
                 "v_mul_lo_u32 %7, %7, %[v]\n"
                 "v_mul_lo_u32 %0, %0, %[v]\n"
                 "v_mul_lo_u32 %1, %1, %[v]\n"
                 "v_mul_lo_u32 %2, %2, %[v]\n"
                 "v_mul_lo_u32 %3, %3, %[v]\n"
                 "v_mul_lo_u32 %4, %4, %[v]\n"
                 "v_mul_lo_u32 %5, %5, %[v]\n"
                 "v_mul_lo_u32 %6, %6, %[v]\n"
                 "v_mul_lo_u32 %7, %7, %[v]\n"
                 "v_mul_lo_u32 %0, %0, %[v]\n"
                 "v_mul_lo_u32 %1, %1, %[v]\n"
                 "v_mul_lo_u32 %2, %2, %[v]\n"
                 "v_mul_lo_u32 %3, %3, %[v]\n"
                 "v_mul_lo_u32 %4, %4, %[v]\n"
                 "v_mul_lo_u32 %5, %5, %[v]\n"
                 "v_mul_lo_u32 %6, %6, %[v]\n"
                 "v_mul_lo_u32 %7, %7, %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_14(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mul_u32_u24 %0, %0, %[v]\n"
                 "v_mul_u32_u24 %1, %1, %[v]\n"
                 "v_mul_u32_u24 %2, %2, %[v]\n"
                 "v_mul_u32_u24 %3, %3, %[v]\n"
                 "v_mul_u32_u24 %4, %4, %[v]\n"
                 "v_mul_u32_u24 %5, %5, %[v]\n"
                 "v_mul_u32_u24 %6, %6, %[v]\n"
                 "v_mul_u32_u24 %7, %7, %[v]\n"
                 "v_mul_u32_u24 %0, %0, %[v]\n"
                 "v_mul_u32_u24 %1, %1, %[v]\n"
                 "v_mul_u32_u24 %2, %2, %[v]\n"
                 "v_mul_u32_u24 %3, %3, %[v]\n"
                 "v_mul_u32_u24 %4, %4, %[v]\n"
                 "v_mul_u32_u24 %5, %5, %[v]\n"
                 "v_mul_u32_u24 %6, %6, %[v]\n"
                 "v_mul_u32_u24 %7, %7, %[v]\n"
                 "v_mul_u32_u24 %0, %0, %[v]\n"
                 "v_mul_u32_u24 %1, %1, %[v]\n"
                 "v_mul_u32_u24 %2, %2, %[v]\n"
                 "v_mul_u32_u24 %3, %3, %[v]\n"
                 "v_mul_u32_u24 %4, %4, %[v]\n"
                 "v_mul_u32_u24 %5, %5, %[v]\n"
                 "v_mul_u32_u24 %6, %6, %[v]\n"
                 "v_mul_u32_u24 %7, %7, %[v]\n"
                 "v_mul_u32_u24 %0, %0, %[v]\n"
                 "v_mul_u32_u24 %1, %1, %[v]\n"
                 "v_mul_u32_u24 %2, %2, %[v]\n"
                 "v_mul_u32_u24 %3, %3, %[v]\n"
                 "v_mul_u32_u24 %4, %4, %[v]\n"
                 "v_mul_u32_u24 %5, %5, %[v]\n"
                 "v_mul_u32_u24 %6, %6, %[v]\n"
                 "v_mul_u32_u24 %7, %7, %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_15(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mad_u32_u24 %0, %0, %[v], %[v]\n"
                 "v_mad_u32_u24 %1, %1, %[v], %[v]\n"
                 "v_mad_u32_u24 %2, %2, %[v], %[v]\n"
                 "v_mad_u32_u24 %3, %3, %[v], %[v]\n"
                 "v_mad_u32_u24 %4, %4, %[v], %[v]\n"
                 "v_mad_u32_u24 %5, %5, %[v], %[v]\n"
                 "v_mad_u32_u24 %6, %6, %[v], %[v]\n"
                 "v_mad_u32_u24 %7, %7, %[v], %[v]\n"
                 "v_mad_u32_u24 %0, %0, %[v], %[v]\n"
                 "v_mad_u32_u24 %1, %1, %[v], %[v]\n"
                 "v_mad_u32_u24 %2, %2, %[v], %[v]\n"
                 "v_mad_u32_u24 %3, %3, %[v], %[v]\n"
                 "v_mad_u32_u24 %4, %4, %[v], %[v]\n"
                 "v_mad_u32_u24 %5, %5, %[v], %[v]\n"
                 "v_mad_u32_u24 %6, %6, %[v], %[v]\n"
                 "v_mad_u32_u24 %7, %7, %[v], %[v]\n"
                 "v_mad_u32_u24 %0, %0, %[v], %[v]\n"
                 "v_mad_u32_u24 %1, %1, %[v], %[v]\n"
                 "v_mad_u32_u24 %2, %2, %[v], %[v]\n"
                 "v_mad_u32_u24 %3, %3, %[v], %[v]\n"
                 "v_mad_u32_u24 %4, %4, %[v], %[v]\n"
                 "v_mad_u32_u24 %5, %5, %[v], %[v]\n"
                 "v_mad_u32_u24 %6, %6, %[v], %[v]\n"
                 "v_mad_u32_u24 %7, %7, %[v], %[v]\n"
                 "v_mad_u32_u24 %0, %0, %[v], %[v]\n"
                 "v_mad_u32_u24 %1, %1, %[v], %[v]\n"
                 "v_mad_u32_u24 %2, %2, %[v], %[v]\n"
                 "v_mad_u32_u24 %3, %3, %[v], %[v]\n"
                 "v_mad_u32_u24 %4, %4, %[v], %[v]\n"
                 "v_mad_u32_u24 %5, %5, %[v], %[v]\n"
                 "v_mad_u32_u24 %6, %6, %[v], %[v]\n"
                 "v_mad_u32_u24 %7, %7, %[v], %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_16(float* out) {
  unsigned long long a0 = threadIdx.x + 0; unsigned long long a1 = threadIdx.x + 1; unsigned long long a2 = threadIdx.x + 2; unsigned long long a3 = threadIdx.x + 3; unsigned long long a4 = threadIdx.x + 4; unsigned long long a5 = threadIdx.x + 5; unsigned long long a6 = threadIdx.x + 6; unsigned long long a7 = threadIdx.x + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mad_u64_u32 %0, vcc, %[v], %[v], %0\n"
                 "v_mad_u64_u32 %1, vcc, %[v], %[v], %1\n"
                 "v_mad_u64_u32 %2, vcc, %[v], %[v], %2\n"
                 "v_mad_u64_u32 %3, vcc, %[v], %[v], %3\n"
                 "v_mad_u64_u32 %4, vcc, %[v], %[v], %4\n"
                 "v_mad_u64_u32 %5, vcc, %[v], %[v], %5\n"
                 "v_mad_u64_u32 %6, vcc, %[v], %[v], %6\n"
                 "v_mad_u64_u32 %7, vcc, %[v], %[v], %7\n"
                 "v_mad_u64_u32 %0, vcc, %[v], %[v], %0\n"
                 "v_mad_u64_u32 %1, vcc, %[v], %[v], %1\n"
                 "v_mad_u64_u32 %2, vcc, %[v], %[v], %2\n"
                 "v_mad_u64_u32 %3, vcc, %[v], %[v], %3\n"
                 "v_mad_u64_u32 %4, vcc, %[v], %[v], %4\n"
                 "v_mad_u64_u32 %5, vcc, %[v], %[v], %5\n"
                 "v_mad_u64_u32 %6, vcc, %[v], %[v], %6\n"
                 "v_mad_u64_u32 %7, vcc, %[v], %[v], %7\n"
                 "v_mad_u64_u32 %0, vcc, %[v], %[v], %0\n"
                 "v_mad_u64_u32 %1, vcc, %[v], %[v], %1\n"
                 "v_mad_u64_u32 %2, vcc, %[v], %[v], %2\n"
                 "v_mad_u64_u32 %3, vcc, %[v], %[v], %3\n"
                 "v_mad_u64_u32 %4, vcc, %[v], %[v], %4\n"
                 "v_mad_u64_u32 %5, vcc, %[v], %[v], %5\n"
                 "v_mad_u64_u32 %6, vcc, %[v], %[v], %6\n"
                 "v_mad_u64_u32 %7, vcc, %[v], %[v], %7\n"
                 "v_mad_u64_u32 %0, vcc, %[v], %[v], %0\n"
                 "v_mad_u64_u32 %1, vcc, %[v], %[v], %1\n"
                 "v_mad_u64_u32 %2, vcc, %[v], %[v], %2\n"
                 "v_mad_u64_u32 %3, vcc, %[v], %[v], %3\n"
                 "v_mad_u64_u32 %4, vcc, %[v], %[v], %4\n"
                 "v_mad_u64_u32 %5, vcc, %[v], %[v], %5\n"
                 "v_mad_u64_u32 %6, vcc, %[v], %[v], %6\n"
                 "v_mad_u64_u32 %7, vcc, %[v], %[v], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7ull && a1 == 7ull && a2 == 7ull && a3 == 7ull && a4 == 7ull && a5 == 7ull && a6 == 7ull && a7 == 7ull) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_17(float* out) {
  unsigned long long a0 = threadIdx.x + 0; unsigned long long a1 = threadIdx.x + 1; unsigned long long a2 = threadIdx.x + 2; unsigned long long a3 = threadIdx.x + 3; unsigned long long a4 = threadIdx.x + 4; unsigned long long a5 = threadIdx.x + 5; unsigned long long a6 = threadIdx.x + 6; unsigned long long a7 = threadIdx.x + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_lshl_add_u64 %0, %0, 0, %0\n"
                 "v_lshl_add_u64 %1, %1, 0, %1\n"
                 "v_lshl_add_u64 %2, %2, 0, %2\n"
                 "v_lshl_add_u64 %3, %3, 0, %3\n"
                 "v_lshl_add_u64 %4, %4, 0, %4\n"
                 "v_lshl_add_u64 %5, %5, 0, %5\n"
                 "v_lshl_add_u64 %6, %6, 0, %6\n"
                 "v_lshl_add_u64 %7, %7, 0, %7\n"
                 "v_lshl_add_u64 %0, %0, 0, %0\n"
                 "v_lshl_add_u64 %1, %1, 0, %1\n"
                 "v_lshl_add_u64 %2, %2, 0, %2\n"
                 "v_lshl_add_u64 %3, %3, 0, %3\n"
                 "v_lshl_add_u64 %4, %4, 0, %4\n"
                 "v_lshl_add_u64 %5, %5, 0, %5\n"
                 "v_lshl_add_u64 %6, %6, 0, %6\n"
                 "v_lshl_add_u64 %7, %7, 0, %7\n"
                 "v_lshl_add_u64 %0, %0, 0, %0\n"
                 "v_lshl_add_u64 %1, %1, 0, %1\n"
                 "v_lshl_add_u64 %2, %2, 0, %2\n"
                 "v_lshl_add_u64 %3, %3, 0, %3\n"
                 "v_lshl_add_u64 %4, %4, 0, %4\n"
                 "v_lshl_add_u64 %5, %5, 0, %5\n"
                 "v_lshl_add_u64 %6, %6, 0, %6\n"
                 "v_lshl_add_u64 %7, %7, 0, %7\n"
                 "v_lshl_add_u64 %0, %0, 0, %0\n"
                 "v_lshl_add_u64 %1, %1, 0, %1\n"
                 "v_lshl_add_u64 %2, %2, 0, %2\n"
                 "v_lshl_add_u64 %3, %3, 0, %3\n"
                 "v_lshl_add_u64 %4, %4, 0, %4\n"
                 "v_lshl_add_u64 %5, %5, 0, %5\n"
                 "v_lshl_add_u64 %6, %6, 0, %6\n"
                 "v_lshl_add_u64 %7, %7, 0, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7ull && a1 == 7ull && a2 == 7ull && a3 == 7ull && a4 == 7ull && a5 == 7ull && a6 == 7ull && a7 == 7ull) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_18(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_exp_f32 %0, %0\n"
                 "v_exp_f32 %1, %1\n"
                 "v_exp_f32 %2, %2\n"
                 "v_exp_f32 %3, %3\n"
                 "v_exp_f32 %4, %4\n"
                 "v_exp_f32 %5, %5\n"
                 "v_exp_f32 %6, %6\n"
                 "v_exp_f32 %7, %7\n"
                 "v_exp_f32 %0, %0\n"
                 "v_exp_f32 %1, %1\n"
                 "v_exp_f32 %2, %2\n"
                 "v_exp_f32 %3, %3\n"
                 "v_exp_f32 %4, %4\n"
                 "v_exp_f32 %5, %5\n"
                 "v_exp_f32 %6, %6\n"
                 "v_exp_f32 %7, %7\n"
                 "v_exp_f32 %0, %0\n"
                 "v_exp_f32 %1, %1\n"
                 "v_exp_f32 %2, %2\n"
                 "v_exp_f32 %3, %3\n"
                 "v_exp_f32 %4, %4\n"
                 "v_exp_f32 %5, %5\n"
                 "v_exp_f32 %6, %6\n"
                 "v_exp_f32 %7, %7\n"
                 "v_exp_f32 %0, %0\n"
                 "v_exp_f32 %1, %1\n"
                 "v_exp_f32 %2, %2\n"
                 "v_exp_f32 %3, %3\n"
                 "v_exp_f32 %4, %4\n"
                 "v_exp_f32 %5, %5\n"
                 "v_exp_f32 %6, %6\n"
                 "v_exp_f32 %7, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_19(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_rcp_f32 %0, %0\n"
                 "v_rcp_f32 %1, %1\n"
                 "v_rcp_f32 %2, %2\n"
                 "v_rcp_f32 %3, %3\n"
                 "v_rcp_f32 %4, %4\n"
                 "v_rcp_f32 %5, %5\n"
                 "v_rcp_f32 %6, %6\n"
                 "v_rcp_f32 %7, %7\n"
                 "v_rcp_f32 %0, %0\n"
                 "v_rcp_f32 %1, %1\n"
                 "v_rcp_f32 %2, %2\n"
                 "v_rcp_f32 %3, %3\n"
                 "v_rcp_f32 %4, %4\n"
                 "v_rcp_f32 %5, %5\n"
                 "v_rcp_f32 %6, %6\n"
                 "v_rcp_f32 %7, %7\n"
                 "v_rcp_f32 %0, %0\n"
                 "v_rcp_f32 %1, %1\n"
                 "v_rcp_f32 %2, %2\n"
                 "v_rcp_f32 %3, %3\n"
                 "v_rcp_f32 %4, %4\n"
                 "v_rcp_f32 %5, %5\n"
                 "v_rcp_f32 %6, %6\n"
                 "v_rcp_f32 %7, %7\n"
                 "v_rcp_f32 %0, %0\n"
                 "v_rcp_f32 %1, %1\n"
                 "v_rcp_f32 %2, %2\n"
                 "v_rcp_f32 %3, %3\n"
                 "v_rcp_f32 %4, %4\n"
                 "v_rcp_f32 %5, %5\n"
                 "v_rcp_f32 %6, %6\n"
                 "v_rcp_f32 %7, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_20(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cndmask_b32 %0, %[b], %0, vcc\n"
                 "v_cndmask_b32 %1, %[b], %1, vcc\n"
                 "v_cndmask_b32 %2, %[b], %2, vcc\n"
                 "v_cndmask_b32 %3, %[b], %3, vcc\n"
                 "v_cndmask_b32 %4, %[b], %4, vcc\n"
                 "v_cndmask_b32 %5, %[b], %5, vcc\n"
                 "v_cndmask_b32 %6, %[b], %6, vcc\n"
                 "v_cndmask_b32 %7, %[b], %7, vcc\n"
                 "v_cndmask_b32 %0, %[b], %0, vcc\n"
                 "v_cndmask_b32 %1, %[b], %1, vcc\n"
                 "v_cndmask_b32 %2, %[b], %2, vcc\n"
                 "v_cndmask_b32 %3, %[b], %3, vcc\n"
                 "v_cndmask_b32 %4, %[b], %4, vcc\n"
                 "v_cndmask_b32 %5, %[b], %5, vcc\n"
                 "v_cndmask_b32 %6, %[b], %6, vcc\n"
                 "v_cndmask_b32 %7, %[b], %7, vcc\n"
                 "v_cndmask_b32 %0, %[b], %0, vcc\n"
                 "v_cndmask_b32 %1, %[b], %1, vcc\n"
                 "v_cndmask_b32 %2, %[b], %2, vcc\n"
                 "v_cndmask_b32 %3, %[b], %3, vcc\n"
                 "v_cndmask_b32 %4, %[b], %4, vcc\n"
                 "v_cndmask_b32 %5, %[b], %5, vcc\n"
                 "v_cndmask_b32 %6, %[b], %6, vcc\n"
                 "v_cndmask_b32 %7, %[b], %7, vcc\n"
                 "v_cndmask_b32 %0, %[b], %0, vcc\n"
                 "v_cndmask_b32 %1, %[b], %1, vcc\n"
                 "v_cndmask_b32 %2, %[b], %2, vcc\n"
                 "v_cndmask_b32 %3, %[b], %3, vcc\n"
                 "v_cndmask_b32 %4, %[b], %4, vcc\n"
                 "v_cndmask_b32 %5, %[b], %5, vcc\n"
                 "v_cndmask_b32 %6, %[b], %6, vcc\n"
                 "v_cndmask_b32 %7, %[b], %7, vcc\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_21(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cndmask_b32_e64 %0, %[b], %0, s[20:21]\n"
                 "v_cndmask_b32_e64 %1, %[b], %1, s[20:21]\n"
                 "v_cndmask_b32_e64 %2, %[b], %2, s[20:21]\n"
                 "v_cndmask_b32_e64 %3, %[b], %3, s[20:21]\n"
                 "v_cndmask_b32_e64 %4, %[b], %4, s[20:21]\n"
                 "v_cndmask_b32_e64 %5, %[b], %5, s[20:21]\n"
                 "v_cndmask_b32_e64 %6, %[b], %6, s[20:21]\n"
                 "v_cndmask_b32_e64 %7, %[b], %7, s[20:21]\n"
                 "v_cndmask_b32_e64 %0, %[b], %0, s[20:21]\n"
                 "v_cndmask_b32_e64 %1, %[b], %1, s[20:21]\n"
                 "v_cndmask_b32_e64 %2, %[b], %2, s[20:21]\n"
                 "v_cndmask_b32_e64 %3, %[b], %3, s[20:21]\n"
                 "v_cndmask_b32_e64 %4, %[b], %4, s[20:21]\n"
                 "v_cndmask_b32_e64 %5, %[b], %5, s[20:21]\n"
                 "v_cndmask_b32_e64 %6, %[b], %6, s[20:21]\n"
                 "v_cndmask_b32_e64 %7, %[b], %7, s[20:21]\n"
                 "v_cndmask_b32_e64 %0, %[b], %0, s[20:21]\n"
                 "v_cndmask_b32_e64 %1, %[b], %1, s[20:21]\n"
                 "v_cndmask_b32_e64 %2, %[b], %2, s[20:21]\n"
                 "v_cndmask_b32_e64 %3, %[b], %3, s[20:21]\n"
                 "v_cndmask_b32_e64 %4, %[b], %4, s[20:21]\n"
                 "v_cndmask_b32_e64 %5, %[b], %5, s[20:21]\n"
                 "v_cndmask_b32_e64 %6, %[b], %6, s[20:21]\n"
                 "v_cndmask_b32_e64 %7, %[b], %7, s[20:21]\n"
                 "v_cndmask_b32_e64 %0, %[b], %0, s[20:21]\n"
                 "v_cndmask_b32_e64 %1, %[b], %1, s[20:21]\n"
                 "v_cndmask_b32_e64 %2, %[b], %2, s[20:21]\n"
                 "v_cndmask_b32_e64 %3, %[b], %3, s[20:21]\n"
                 "v_cndmask_b32_e64 %4, %[b], %4, s[20:21]\n"
                 "v_cndmask_b32_e64 %5, %[b], %5, s[20:21]\n"
                 "v_cndmask_b32_e64 %6, %[b], %6, s[20:21]\n"
                 "v_cndmask_b32_e64 %7, %[b], %7, s[20:21]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_22(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_max3_f32 %0, %[b], %[c], %0\n"
                 "v_max3_f32 %1, %[b], %[c], %1\n"
                 "v_max3_f32 %2, %[b], %[c], %2\n"
                 "v_max3_f32 %3, %[b], %[c], %3\n"
                 "v_max3_f32 %4, %[b], %[c], %4\n"
                 "v_max3_f32 %5, %[b], %[c], %5\n"
                 "v_max3_f32 %6, %[b], %[c], %6\n"
                 "v_max3_f32 %7, %[b], %[c], %7\n"
                 "v_max3_f32 %0, %[b], %[c], %0\n"
                 "v_max3_f32 %1, %[b], %[c], %1\n"
                 "v_max3_f32 %2, %[b], %[c], %2\n"
                 "v_max3_f32 %3, %[b], %[c], %3\n"
                 "v_max3_f32 %4, %[b], %[c], %4\n"
                 "v_max3_f32 %5, %[b], %[c], %5\n"
                 "v_max3_f32 %6, %[b], %[c], %6\n"
                 "v_max3_f32 %7, %[b], %[c], %7\n"
                 "v_max3_f32 %0, %[b], %[c], %0\n"
                 "v_max3_f32 %1, %[b], %[c], %1\n"
                 "v_max3_f32 %2, %[b], %[c], %2\n"
                 "v_max3_f32 %3, %[b], %[c], %3\n"
                 "v_max3_f32 %4, %[b], %[c], %4\n"
                 "v_max3_f32 %5, %[b], %[c], %5\n"
                 "v_max3_f32 %6, %[b], %[c], %6\n"
                 "v_max3_f32 %7, %[b], %[c], %7\n"
                 "v_max3_f32 %0, %[b], %[c], %0\n"
                 "v_max3_f32 %1, %[b], %[c], %1\n"
                 "v_max3_f32 %2, %[b], %[c], %2\n"
                 "v_max3_f32 %3, %[b], %[c], %3\n"
                 "v_max3_f32 %4, %[b], %[c], %4\n"
                 "v_max3_f32 %5, %[b], %[c], %5\n"
                 "v_max3_f32 %6, %[b], %[c], %6\n"
                 "v_max3_f32 %7, %[b], %[c], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_23(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_24(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mov_b32_dpp %0, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %2, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %3, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %5, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %6, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %7, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %0, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %2, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %3, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %5, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %6, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %7, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %0, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %2, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %3, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %5, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %6, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %7, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %0, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %2, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %3, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %4, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %5, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %6, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %7, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_25(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cvt_pk_bf16_f32 %0, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %1, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %2, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %3, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %4, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %5, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %6, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %7, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %0, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %1, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %2, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %3, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %4, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %5, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %6, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %7, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %0, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %1, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %2, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %3, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %4, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %5, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %6, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %7, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %0, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %1, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %2, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %3, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %4, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %5, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %6, %[b], %[c]\n"
                 "v_cvt_pk_bf16_f32 %7, %[b], %[c]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_26(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cvt_f32_bf16 %0, %[b]\n"
                 "v_cvt_f32_bf16 %1, %[b]\n"
                 "v_cvt_f32_bf16 %2, %[b]\n"
                 "v_cvt_f32_bf16 %3, %[b]\n"
                 "v_cvt_f32_bf16 %4, %[b]\n"
                 "v_cvt_f32_bf16 %5, %[b]\n"
                 "v_cvt_f32_bf16 %6, %[b]\n"
                 "v_cvt_f32_bf16 %7, %[b]\n"
                 "v_cvt_f32_bf16 %0, %[b]\n"
                 "v_cvt_f32_bf16 %1, %[b]\n"
                 "v_cvt_f32_bf16 %2, %[b]\n"
                 "v_cvt_f32_bf16 %3, %[b]\n"
                 "v_cvt_f32_bf16 %4, %[b]\n"
                 "v_cvt_f32_bf16 %5, %[b]\n"
                 "v_cvt_f32_bf16 %6, %[b]\n"
                 "v_cvt_f32_bf16 %7, %[b]\n"
                 "v_cvt_f32_bf16 %0, %[b]\n"
                 "v_cvt_f32_bf16 %1, %[b]\n"
                 "v_cvt_f32_bf16 %2, %[b]\n"
                 "v_cvt_f32_bf16 %3, %[b]\n"
                 "v_cvt_f32_bf16 %4, %[b]\n"
                 "v_cvt_f32_bf16 %5, %[b]\n"
                 "v_cvt_f32_bf16 %6, %[b]\n"
                 "v_cvt_f32_bf16 %7, %[b]\n"
                 "v_cvt_f32_bf16 %0, %[b]\n"
                 "v_cvt_f32_bf16 %1, %[b]\n"
                 "v_cvt_f32_bf16 %2, %[b]\n"
                 "v_cvt_f32_bf16 %3, %[b]\n"
                 "v_cvt_f32_bf16 %4, %[b]\n"
                 "v_cvt_f32_bf16 %5, %[b]\n"
                 "v_cvt_f32_bf16 %6, %[b]\n"
                 "v_cvt_f32_bf16 %7, %[b]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_27(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cvt_f32_bf16_sdwa %0, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %1, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %2, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %3, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %4, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %5, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %6, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %7, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %0, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %1, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %2, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %3, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %4, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %5, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %6, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %7, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %0, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %1, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %2, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %3, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %4, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %5, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %6, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %7, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %0, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %1, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %2, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %3, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %4, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %5, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %6, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 "v_cvt_f32_bf16_sdwa %7, %[b] dst_sel:DWORD src0_sel:WORD_1\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_28(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_fma_mix_f32 %0, %[b], %[c], %0 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %1, %[b], %[c], %1 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %2, %[b], %[c], %2 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %3, %[b], %[c], %3 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %4, %[b], %[c], %4 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %5, %[b], %[c], %5 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %6, %[b], %[c], %6 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %7, %[b], %[c], %7 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %0, %[b], %[c], %0 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %1, %[b], %[c], %1 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %2, %[b], %[c], %2 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %3, %[b], %[c], %3 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %4, %[b], %[c], %4 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %5, %[b], %[c], %5 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %6, %[b], %[c], %6 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %7, %[b], %[c], %7 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %0, %[b], %[c], %0 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %1, %[b], %[c], %1 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %2, %[b], %[c], %2 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %3, %[b], %[c], %3 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %4, %[b], %[c], %4 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %5, %[b], %[c], %5 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %6, %[b], %[c], %6 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %7, %[b], %[c], %7 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %0, %[b], %[c], %0 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %1, %[b], %[c], %1 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %2, %[b], %[c], %2 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %3, %[b], %[c], %3 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %4, %[b], %[c], %4 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %5, %[b], %[c], %5 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %6, %[b], %[c], %6 op_sel_hi:[1,1,0]\n"
                 "v_fma_mix_f32 %7, %[b], %[c], %7 op_sel_hi:[1,1,0]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_29(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_dot2c_f32_bf16 %0, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %1, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %2, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %3, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %4, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %5, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %6, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %7, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %0, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %1, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %2, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %3, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %4, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %5, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %6, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %7, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %0, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %1, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %2, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %3, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %4, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %5, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %6, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %7, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %0, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %1, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %2, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %3, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %4, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %5, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %6, %[b], %[c]\n"
                 "v_dot2c_f32_bf16 %7, %[b], %[c]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_30(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_dot2_f32_bf16 %0, %[b], %[c], %0\n"
                 "v_dot2_f32_bf16 %1, %[b], %[c], %1\n"
                 "v_dot2_f32_bf16 %2, %[b], %[c], %2\n"
                 "v_dot2_f32_bf16 %3, %[b], %[c], %3\n"
                 "v_dot2_f32_bf16 %4, %[b], %[c], %4\n"
                 "v_dot2_f32_bf16 %5, %[b], %[c], %5\n"
                 "v_dot2_f32_bf16 %6, %[b], %[c], %6\n"
                 "v_dot2_f32_bf16 %7, %[b], %[c], %7\n"
                 "v_dot2_f32_bf16 %0, %[b], %[c], %0\n"
                 "v_dot2_f32_bf16 %1, %[b], %[c], %1\n"
                 "v_dot2_f32_bf16 %2, %[b], %[c], %2\n"
                 "v_dot2_f32_bf16 %3, %[b], %[c], %3\n"
                 "v_dot2_f32_bf16 %4, %[b], %[c], %4\n"
                 "v_dot2_f32_bf16 %5, %[b], %[c], %5\n"
                 "v_dot2_f32_bf16 %6, %[b], %[c], %6\n"
                 "v_dot2_f32_bf16 %7, %[b], %[c], %7\n"
                 "v_dot2_f32_bf16 %0, %[b], %[c], %0\n"
                 "v_dot2_f32_bf16 %1, %[b], %[c], %1\n"
                 "v_dot2_f32_bf16 %2, %[b], %[c], %2\n"
                 "v_dot2_f32_bf16 %3, %[b], %[c], %3\n"
                 "v_dot2_f32_bf16 %4, %[b], %[c], %4\n"
                 "v_dot2_f32_bf16 %5, %[b], %[c], %5\n"
                 "v_dot2_f32_bf16 %6, %[b], %[c], %6\n"
                 "v_dot2_f32_bf16 %7, %[b], %[c], %7\n"
                 "v_dot2_f32_bf16 %0, %[b], %[c], %0\n"
                 "v_dot2_f32_bf16 %1, %[b], %[c], %1\n"
                 "v_dot2_f32_bf16 %2, %[b], %[c], %2\n"
                 "v_dot2_f32_bf16 %3, %[b], %[c], %3\n"
                 "v_dot2_f32_bf16 %4, %[b], %[c], %4\n"
                 "v_dot2_f32_bf16 %5, %[b], %[c], %5\n"
                 "v_dot2_f32_bf16 %6, %[b], %[c], %6\n"
                 "v_dot2_f32_bf16 %7, %[b], %[c], %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_31(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_perm_b32 %0, %0, %[v], %[v]\n"
                 "v_perm_b32 %1, %1, %[v], %[v]\n"
                 "v_perm_b32 %2, %2, %[v], %[v]\n"
                 "v_perm_b32 %3, %3, %[v], %[v]\n"
                 "v_perm_b32 %4, %4, %[v], %[v]\n"
                 "v_perm_b32 %5, %5, %[v], %[v]\n"
                 "v_perm_b32 %6, %6, %[v], %[v]\n"
                 "v_perm_b32 %7, %7, %[v], %[v]\n"
                 "v_perm_b32 %0, %0, %[v], %[v]\n"
                 "v_perm_b32 %1, %1, %[v], %[v]\n"
                 "v_perm_b32 %2, %2, %[v], %[v]\n"
                 "v_perm_b32 %3, %3, %[v], %[v]\n"
                 "v_perm_b32 %4, %4, %[v], %[v]\n"
                 "v_perm_b32 %5, %5, %[v], %[v]\n"
                 "v_perm_b32 %6, %6, %[v], %[v]\n"
                 "v_perm_b32 %7, %7, %[v], %[v]\n"
                 "v_perm_b32 %0, %0, %[v], %[v]\n"
                 "v_perm_b32 %1, %1, %[v], %[v]\n"
                 "v_perm_b32 %2, %2, %[v], %[v]\n"
                 "v_perm_b32 %3, %3, %[v], %[v]\n"
                 "v_perm_b32 %4, %4, %[v], %[v]\n"
                 "v_perm_b32 %5, %5, %[v], %[v]\n"
                 "v_perm_b32 %6, %6, %[v], %[v]\n"
                 "v_perm_b32 %7, %7, %[v], %[v]\n"
                 "v_perm_b32 %0, %0, %[v], %[v]\n"
                 "v_perm_b32 %1, %1, %[v], %[v]\n"
                 "v_perm_b32 %2, %2, %[v], %[v]\n"
                 "v_perm_b32 %3, %3, %[v], %[v]\n"
                 "v_perm_b32 %4, %4, %[v], %[v]\n"
                 "v_perm_b32 %5, %5, %[v], %[v]\n"
                 "v_perm_b32 %6, %6, %[v], %[v]\n"
                 "v_perm_b32 %7, %7, %[v], %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_32(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cmp_lt_u32 vcc, %0, %[v]\n"
                 "v_cmp_lt_u32 vcc, %1, %[v]\n"
                 "v_cmp_lt_u32 vcc, %2, %[v]\n"
                 "v_cmp_lt_u32 vcc, %3, %[v]\n"
                 "v_cmp_lt_u32 vcc, %4, %[v]\n"
                 "v_cmp_lt_u32 vcc, %5, %[v]\n"
                 "v_cmp_lt_u32 vcc, %6, %[v]\n"
                 "v_cmp_lt_u32 vcc, %7, %[v]\n"
                 "v_cmp_lt_u32 vcc, %0, %[v]\n"
                 "v_cmp_lt_u32 vcc, %1, %[v]\n"
                 "v_cmp_lt_u32 vcc, %2, %[v]\n"
                 "v_cmp_lt_u32 vcc, %3, %[v]\n"
                 "v_cmp_lt_u32 vcc, %4, %[v]\n"
                 "v_cmp_lt_u32 vcc, %5, %[v]\n"
                 "v_cmp_lt_u32 vcc, %6, %[v]\n"
                 "v_cmp_lt_u32 vcc, %7, %[v]\n"
                 "v_cmp_lt_u32 vcc, %0, %[v]\n"
                 "v_cmp_lt_u32 vcc, %1, %[v]\n"
                 "v_cmp_lt_u32 vcc, %2, %[v]\n"
                 "v_cmp_lt_u32 vcc, %3, %[v]\n"
                 "v_cmp_lt_u32 vcc, %4, %[v]\n"
                 "v_cmp_lt_u32 vcc, %5, %[v]\n"
                 "v_cmp_lt_u32 vcc, %6, %[v]\n"
                 "v_cmp_lt_u32 vcc, %7, %[v]\n"
                 "v_cmp_lt_u32 vcc, %0, %[v]\n"
                 "v_cmp_lt_u32 vcc, %1, %[v]\n"
                 "v_cmp_lt_u32 vcc, %2, %[v]\n"
                 "v_cmp_lt_u32 vcc, %3, %[v]\n"
                 "v_cmp_lt_u32 vcc, %4, %[v]\n"
                 "v_cmp_lt_u32 vcc, %5, %[v]\n"
                 "v_cmp_lt_u32 vcc, %6, %[v]\n"
                 "v_cmp_lt_u32 vcc, %7, %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_33(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %1, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %2, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %3, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %4, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %5, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %6, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %7, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %0, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %1, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %2, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %3, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %4, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %5, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %6, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %7, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %0, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %1, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %2, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %3, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %4, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %5, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %6, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %7, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %0, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %1, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %2, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %3, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %4, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %5, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %6, %[b]\n"
                 "v_cmp_lt_f32_e64 s[20:21], %7, %[b]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_34(float* out) {
  float a0 = threadIdx.x * 1e-3f + 0; float a1 = threadIdx.x * 1e-3f + 1; float a2 = threadIdx.x * 1e-3f + 2; float a3 = threadIdx.x * 1e-3f + 3; float a4 = threadIdx.x * 1e-3f + 4; float a5 = threadIdx.x * 1e-3f + 5; float a6 = threadIdx.x * 1e-3f + 6; float a7 = threadIdx.x * 1e-3f + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_mov_b32 %0, %[b]\n"
                 "v_mov_b32 %1, %[b]\n"
                 "v_mov_b32 %2, %[b]\n"
                 "v_mov_b32 %3, %[b]\n"
                 "v_mov_b32 %4, %[b]\n"
                 "v_mov_b32 %5, %[b]\n"
                 "v_mov_b32 %6, %[b]\n"
                 "v_mov_b32 %7, %[b]\n"
                 "v_mov_b32 %0, %[b]\n"
                 "v_mov_b32 %1, %[b]\n"
                 "v_mov_b32 %2, %[b]\n"
                 "v_mov_b32 %3, %[b]\n"
                 "v_mov_b32 %4, %[b]\n"
                 "v_mov_b32 %5, %[b]\n"
                 "v_mov_b32 %6, %[b]\n"
                 "v_mov_b32 %7, %[b]\n"
                 "v_mov_b32 %0, %[b]\n"
                 "v_mov_b32 %1, %[b]\n"
                 "v_mov_b32 %2, %[b]\n"
                 "v_mov_b32 %3, %[b]\n"
                 "v_mov_b32 %4, %[b]\n"
                 "v_mov_b32 %5, %[b]\n"
                 "v_mov_b32 %6, %[b]\n"
                 "v_mov_b32 %7, %[b]\n"
                 "v_mov_b32 %0, %[b]\n"
                 "v_mov_b32 %1, %[b]\n"
                 "v_mov_b32 %2, %[b]\n"
                 "v_mov_b32 %3, %[b]\n"
                 "v_mov_b32 %4, %[b]\n"
                 "v_mov_b32 %5, %[b]\n"
                 "v_mov_b32 %6, %[b]\n"
                 "v_mov_b32 %7, %[b]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 12345.f && a1 == 12345.f && a2 == 12345.f && a3 == 12345.f && a4 == 12345.f && a5 == 12345.f && a6 == 12345.f && a7 == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_35(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_bfe_u32 %0, %0, 3, 5\n"
                 "v_bfe_u32 %1, %1, 3, 5\n"
                 "v_bfe_u32 %2, %2, 3, 5\n"
                 "v_bfe_u32 %3, %3, 3, 5\n"
                 "v_bfe_u32 %4, %4, 3, 5\n"
                 "v_bfe_u32 %5, %5, 3, 5\n"
                 "v_bfe_u32 %6, %6, 3, 5\n"
                 "v_bfe_u32 %7, %7, 3, 5\n"
                 "v_bfe_u32 %0, %0, 3, 5\n"
                 "v_bfe_u32 %1, %1, 3, 5\n"
                 "v_bfe_u32 %2, %2, 3, 5\n"
                 "v_bfe_u32 %3, %3, 3, 5\n"
                 "v_bfe_u32 %4, %4, 3, 5\n"
                 "v_bfe_u32 %5, %5, 3, 5\n"
                 "v_bfe_u32 %6, %6, 3, 5\n"
                 "v_bfe_u32 %7, %7, 3, 5\n"
                 "v_bfe_u32 %0, %0, 3, 5\n"
                 "v_bfe_u32 %1, %1, 3, 5\n"
                 "v_bfe_u32 %2, %2, 3, 5\n"
                 "v_bfe_u32 %3, %3, 3, 5\n"
                 "v_bfe_u32 %4, %4, 3, 5\n"
                 "v_bfe_u32 %5, %5, 3, 5\n"
                 "v_bfe_u32 %6, %6, 3, 5\n"
                 "v_bfe_u32 %7, %7, 3, 5\n"
                 "v_bfe_u32 %0, %0, 3, 5\n"
                 "v_bfe_u32 %1, %1, 3, 5\n"
                 "v_bfe_u32 %2, %2, 3, 5\n"
                 "v_bfe_u32 %3, %3, 3, 5\n"
                 "v_bfe_u32 %4, %4, 3, 5\n"
                 "v_bfe_u32 %5, %5, 3, 5\n"
                 "v_bfe_u32 %6, %6, 3, 5\n"
                 "v_bfe_u32 %7, %7, 3, 5\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_36(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_and_or_b32 %0, %0, %[v], %[v]\n"
                 "v_and_or_b32 %1, %1, %[v], %[v]\n"
                 "v_and_or_b32 %2, %2, %[v], %[v]\n"
                 "v_and_or_b32 %3, %3, %[v], %[v]\n"
                 "v_and_or_b32 %4, %4, %[v], %[v]\n"
                 "v_and_or_b32 %5, %5, %[v], %[v]\n"
                 "v_and_or_b32 %6, %6, %[v], %[v]\n"
                 "v_and_or_b32 %7, %7, %[v], %[v]\n"
                 "v_and_or_b32 %0, %0, %[v], %[v]\n"
                 "v_and_or_b32 %1, %1, %[v], %[v]\n"
                 "v_and_or_b32 %2, %2, %[v], %[v]\n"
                 "v_and_or_b32 %3, %3, %[v], %[v]\n"
                 "v_and_or_b32 %4, %4, %[v], %[v]\n"
                 "v_and_or_b32 %5, %5, %[v], %[v]\n"
                 "v_and_or_b32 %6, %6, %[v], %[v]\n"
                 "v_and_or_b32 %7, %7, %[v], %[v]\n"
                 "v_and_or_b32 %0, %0, %[v], %[v]\n"
                 "v_and_or_b32 %1, %1, %[v], %[v]\n"
                 "v_and_or_b32 %2, %2, %[v], %[v]\n"
                 "v_and_or_b32 %3, %3, %[v], %[v]\n"
                 "v_and_or_b32 %4, %4, %[v], %[v]\n"
                 "v_and_or_b32 %5, %5, %[v], %[v]\n"
                 "v_and_or_b32 %6, %6, %[v], %[v]\n"
                 "v_and_or_b32 %7, %7, %[v], %[v]\n"
                 "v_and_or_b32 %0, %0, %[v], %[v]\n"
                 "v_and_or_b32 %1, %1, %[v], %[v]\n"
                 "v_and_or_b32 %2, %2, %[v], %[v]\n"
                 "v_and_or_b32 %3, %3, %[v], %[v]\n"
                 "v_and_or_b32 %4, %4, %[v], %[v]\n"
                 "v_and_or_b32 %5, %5, %[v], %[v]\n"
                 "v_and_or_b32 %6, %6, %[v], %[v]\n"
                 "v_and_or_b32 %7, %7, %[v], %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_37(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_lshrrev_b32 %0, 13, %0\n"
                 "v_lshrrev_b32 %1, 13, %1\n"
                 "v_lshrrev_b32 %2, 13, %2\n"
                 "v_lshrrev_b32 %3, 13, %3\n"
                 "v_lshrrev_b32 %4, 13, %4\n"
                 "v_lshrrev_b32 %5, 13, %5\n"
                 "v_lshrrev_b32 %6, 13, %6\n"
                 "v_lshrrev_b32 %7, 13, %7\n"
                 "v_lshrrev_b32 %0, 13, %0\n"
                 "v_lshrrev_b32 %1, 13, %1\n"
                 "v_lshrrev_b32 %2, 13, %2\n"
                 "v_lshrrev_b32 %3, 13, %3\n"
                 "v_lshrrev_b32 %4, 13, %4\n"
                 "v_lshrrev_b32 %5, 13, %5\n"
                 "v_lshrrev_b32 %6, 13, %6\n"
                 "v_lshrrev_b32 %7, 13, %7\n"
                 "v_lshrrev_b32 %0, 13, %0\n"
                 "v_lshrrev_b32 %1, 13, %1\n"
                 "v_lshrrev_b32 %2, 13, %2\n"
                 "v_lshrrev_b32 %3, 13, %3\n"
                 "v_lshrrev_b32 %4, 13, %4\n"
                 "v_lshrrev_b32 %5, 13, %5\n"
                 "v_lshrrev_b32 %6, 13, %6\n"
                 "v_lshrrev_b32 %7, 13, %7\n"
                 "v_lshrrev_b32 %0, 13, %0\n"
                 "v_lshrrev_b32 %1, 13, %1\n"
                 "v_lshrrev_b32 %2, 13, %2\n"
                 "v_lshrrev_b32 %3, 13, %3\n"
                 "v_lshrrev_b32 %4, 13, %4\n"
                 "v_lshrrev_b32 %5, 13, %5\n"
                 "v_lshrrev_b32 %6, 13, %6\n"
                 "v_lshrrev_b32 %7, 13, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void probe_38(float* out) {
  unsigned a0 = threadIdx.x * 2654435761u + 0; unsigned a1 = threadIdx.x * 2654435761u + 1; unsigned a2 = threadIdx.x * 2654435761u + 2; unsigned a3 = threadIdx.x * 2654435761u + 3; unsigned a4 = threadIdx.x * 2654435761u + 4; unsigned a5 = threadIdx.x * 2654435761u + 5; unsigned a6 = threadIdx.x * 2654435761u + 6; unsigned a7 = threadIdx.x * 2654435761u + 7;
  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;
  for (int i = 0; i < REPS; ++i) {
    asm volatile("v_xad_u32 %0, %0, %[v], %[v]\n"
                 "v_xad_u32 %1, %1, %[v], %[v]\n"
                 "v_xad_u32 %2, %2, %[v], %[v]\n"
                 "v_xad_u32 %3, %3, %[v], %[v]\n"
                 "v_xad_u32 %4, %4, %[v], %[v]\n"
                 "v_xad_u32 %5, %5, %[v], %[v]\n"
                 "v_xad_u32 %6, %6, %[v], %[v]\n"
                 "v_xad_u32 %7, %7, %[v], %[v]\n"
                 "v_xad_u32 %0, %0, %[v], %[v]\n"
                 "v_xad_u32 %1, %1, %[v], %[v]\n"
                 "v_xad_u32 %2, %2, %[v], %[v]\n"
                 "v_xad_u32 %3, %3, %[v], %[v]\n"
                 "v_xad_u32 %4, %4, %[v], %[v]\n"
                 "v_xad_u32 %5, %5, %[v], %[v]\n"
                 "v_xad_u32 %6, %6, %[v], %[v]\n"
                 "v_xad_u32 %7, %7, %[v], %[v]\n"
                 "v_xad_u32 %0, %0, %[v], %[v]\n"
                 "v_xad_u32 %1, %1, %[v], %[v]\n"
                 "v_xad_u32 %2, %2, %[v], %[v]\n"
                 "v_xad_u32 %3, %3, %[v], %[v]\n"
                 "v_xad_u32 %4, %4, %[v], %[v]\n"
                 "v_xad_u32 %5, %5, %[v], %[v]\n"
                 "v_xad_u32 %6, %6, %[v], %[v]\n"
                 "v_xad_u32 %7, %7, %[v], %[v]\n"
                 "v_xad_u32 %0, %0, %[v], %[v]\n"
                 "v_xad_u32 %1, %1, %[v], %[v]\n"
                 "v_xad_u32 %2, %2, %[v], %[v]\n"
                 "v_xad_u32 %3, %3, %[v], %[v]\n"
                 "v_xad_u32 %4, %4, %[v], %[v]\n"
                 "v_xad_u32 %5, %5, %[v], %[v]\n"
                 "v_xad_u32 %6, %6, %[v], %[v]\n"
                 "v_xad_u32 %7, %7, %[v], %[v]\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");
  }
  if (a0 == 7u && a1 == 7u && a2 == 7u && a3 == 7u && a4 == 7u && a5 == 7u && a6 == 7u && a7 == 7u) out[0] = 1.f;
}

typedef void (*kern_t)(float*);
struct Entry { const char* name; kern_t fn; };

int main() {
  Entry es[] = {{"v_fma_f32", probe_0}, {"v_fma_f32 |abs|", probe_1}, {"v_add_f32", probe_2}, {"v_mul_f32", probe_3}, {"v_max_f32", probe_4}, {"v_pk_fma_f32", probe_5}, {"v_pk_add_f32", probe_6}, {"v_pk_mul_f32", probe_7}, {"v_and_b32", probe_8}, {"v_lshlrev_b32", probe_9}, {"v_xor_b32", probe_10}, {"v_add_u32", probe_11}, {"v_bitop3_b32", probe_12}, {"v_mul_lo_u32", probe_13}, {"v_mul_u32_u24", probe_14}, {"v_mad_u32_u24", probe_15}, {"v_mad_u64_u32", probe_16}, {"v_lshl_add_u64", probe_17}, {"v_exp_f32", probe_18}, {"v_rcp_f32", probe_19}, {"v_cndmask_b32 vcc", probe_20}, {"v_cndmask_b32 sgpr", probe_21}, {"v_max3_f32", probe_22}, {"v_add_f32_dpp", probe_23}, {"v_mov_b32_dpp", probe_24}, {"v_cvt_pk_bf16_f32", probe_25}, {"v_cvt_f32_bf16", probe_26}, {"v_cvt_f32_bf16 sdwa hi", probe_27}, {"v_fma_mix_f32", probe_28}, {"v_dot2c_f32_bf16", probe_29}, {"v_dot2_f32_bf16", probe_30}, {"v_perm_b32", probe_31}, {"v_cmp_lt_u32", probe_32}, {"v_cmp_lt_f32 sgpr", probe_33}, {"v_mov_b32", probe_34}, {"v_bfe_u32", probe_35}, {"v_and_or_b32", probe_36}, {"v_lshrrev_b32", probe_37}, {"v_xad_u32", probe_38}};

  float* out;
  if (hipMalloc(&out, 64) != hipSuccess) return 1;
  hipEvent_t t0, t1;
  (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  const double clk = 2.4e9;
  printf("%-24s %8s %8s %8s %8s   (SIMD cycles per wave-instruction at N waves/SIMD)\n", "instruction", "1w", "2w", "4w", "8w");
  for (auto& e : es) {
    printf("%-24s", e.name);
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 256 * wps;                       // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(t0);
      for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out);
      (void)hipEventRecord(t1);
      (void)hipEventSynchronize(t1);
      float ms; (void)hipEventElapsedTime(&ms, t0, t1);
      const double per_simd_instr = 3.0 * REPS * 32.0 * wps;   // wave-instructions issued on one SIMD
      printf(" %8.2f", ms * 1e-3 * clk / per_simd_instr);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
