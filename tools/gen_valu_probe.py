#!/usr/bin/env python3
"""Generator of the VALU issue-cost probe (profiles/r02_valu_probe.txt was measured with its output).

    python tools/gen_valu_probe.py > build/valu_probe.hip
    hipcc -O3 --offload-arch=gfx950 build/valu_probe.hip -o build/valu_probe && build/valu_probe

One kernel per row of PROBES: REPS x 32 copies of ONE instruction per wave, cycling over 8 independent destination
registers (%0..%7); kinds: f = float, p = float2 (packed), u = unsigned, q = unsigned long long."""
import sys

PROBES = [
    ("v_fma_f32", "f", "v_fma_f32 %0, %[b], %[c], %0"),
    ("v_fma_f32 |abs|", "f", "v_fma_f32 %0, %[b], |%[c]|, %0"),
    ("v_add_f32", "f", "v_add_f32 %0, %[b], %0"),
    ("v_mul_f32", "f", "v_mul_f32 %0, %[b], %0"),
    ("v_max_f32", "f", "v_max_f32 %0, %[b], %0"),
    ("v_pk_fma_f32", "p", "v_pk_fma_f32 %0, %[q], %[r], %0"),
    ("v_pk_add_f32", "p", "v_pk_add_f32 %0, %[q], %0"),
    ("v_pk_mul_f32", "p", "v_pk_mul_f32 %0, %[q], %0"),
    ("v_and_b32", "u", "v_and_b32 %0, %[v], %0"),
    ("v_lshlrev_b32", "u", "v_lshlrev_b32 %0, 16, %0"),
    ("v_xor_b32", "u", "v_xor_b32 %0, %[v], %0"),
    ("v_add_u32", "u", "v_add_u32 %0, %[v], %0"),
    ("v_bitop3_b32", "u", "v_bitop3_b32 %0, %0, %[v], %[v] bitop3:0x78"),
    ("v_mul_lo_u32", "u", "v_mul_lo_u32 %0, %0, %[v]"),
    ("v_mul_u32_u24", "u", "v_mul_u32_u24 %0, %0, %[v]"),
    ("v_mad_u32_u24", "u", "v_mad_u32_u24 %0, %0, %[v], %[v]"),
    ("v_mad_u64_u32", "q", "v_mad_u64_u32 %0, vcc, %[v], %[v], %0"),
    ("v_lshl_add_u64", "q", "v_lshl_add_u64 %0, %0, 0, %0"),
    ("v_exp_f32", "f", "v_exp_f32 %0, %0"),
    ("v_rcp_f32", "f", "v_rcp_f32 %0, %0"),
    ("v_cndmask_b32 vcc", "f", "v_cndmask_b32 %0, %[b], %0, vcc"),
    ("v_cndmask_b32 sgpr", "f", "v_cndmask_b32_e64 %0, %[b], %0, s[20:21]"),
    ("v_max3_f32", "f", "v_max3_f32 %0, %[b], %[c], %0"),
    ("v_add_f32_dpp", "f", "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("v_mov_b32_dpp", "f", "v_mov_b32_dpp %0, %[b] row_newbcast:1 row_mask:0xf bank_mask:0xf"),
    ("v_cvt_pk_bf16_f32", "u", "v_cvt_pk_bf16_f32 %0, %[b], %[c]"),
    ("v_cvt_f32_bf16", "f", "v_cvt_f32_bf16 %0, %[b]"),
    ("v_cvt_f32_bf16 sdwa hi", "f", "v_cvt_f32_bf16_sdwa %0, %[b] dst_sel:DWORD src0_sel:WORD_1"),
    ("v_fma_mix_f32", "f", "v_fma_mix_f32 %0, %[b], %[c], %0 op_sel_hi:[1,1,0]"),
    ("v_dot2c_f32_bf16", "f", "v_dot2c_f32_bf16 %0, %[b], %[c]"),
    ("v_dot2_f32_bf16", "f", "v_dot2_f32_bf16 %0, %[b], %[c], %0"),
    ("v_perm_b32", "u", "v_perm_b32 %0, %0, %[v], %[v]"),
    ("v_cmp_lt_u32", "u", "v_cmp_lt_u32 vcc, %0, %[v]"),
    ("v_cmp_lt_f32 sgpr", "f", "v_cmp_lt_f32_e64 s[20:21], %0, %[b]"),
    ("v_mov_b32", "f", "v_mov_b32 %0, %[b]"),
    ("v_bfe_u32", "u", "v_bfe_u32 %0, %0, 3, 5"),
    ("v_and_or_b32", "u", "v_and_or_b32 %0, %0, %[v], %[v]"),
    ("v_lshrrev_b32", "u", "v_lshrrev_b32 %0, 13, %0"),
    ("v_xad_u32", "u", "v_xad_u32 %0, %0, %[v], %[v]"),
]

DECL = {
    "f": ("float a{i} = threadIdx.x * 1e-3f + {i};", "a{i} == 12345.f"),
    "p": ("float2 a{i} = {{threadIdx.x * 1e-3f, {i}.f}};", "a{i}.x == 12345.f"),
    "u": ("unsigned a{i} = threadIdx.x * 2654435761u + {i};", "a{i} == 7u"),
    "q": ("unsigned long long a{i} = threadIdx.x + {i};", "a{i} == 7ull"),
}

HEAD = """// Issue-cost probe for the VALU instructions the GATv2 kernels are made of (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o build/valu_probe && build/valu_probe
// Every kernel issues REPS x 32 copies of ONE instruction per wave, cycling over 8 independent destination
// registers; the table is SIMD cycles per wave-instruction at 1, 2, 4 and 8 resident waves per SIMD
// (2.4 GHz assumed: the ratios between rows are what matters).  The kernels in csrc/gatv2_kernels.h are
// VALU-bound, so these prices decide which formulation of an inner loop is cheapest.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REPS 2048
"""

TAIL = """
  float* out;
  if (hipMalloc(&out, 64) != hipSuccess) return 1;
  hipEvent_t t0, t1;
  (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  const double clk = 2.4e9;
  printf("%-24s %8s %8s %8s %8s   (SIMD cycles per wave-instruction at N waves/SIMD)\\n", "instruction", "1w", "2w", "4w", "8w");
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
    printf("\\n");
    fflush(stdout);
  }
  return 0;
}
"""


def kernel(n, kind, ins):
    decl, test = DECL[kind]
    lines = [f"\n__global__ __launch_bounds__(256) void probe_{n}(float* out) {{",
             "  " + " ".join(decl.format(i=i) for i in range(8)),
             "  float b = 1.0001f, c = 0.5f; float2 q = {b, c}, r = {c, b}; unsigned v = 0x9e3779b9u;",
             "  for (int i = 0; i < REPS; ++i) {"]
    body = [ins.replace("%0", f"%{j % 8}") for j in range(32)]
    lines.append('    asm volatile("' + body[0] + '\\n"')
    lines += [f'                 "{b}\\n"' for b in body[1:]]
    lines.append('                 : ' + ", ".join(f'"+v"(a{i})' for i in range(8)))
    lines.append('                 : [b] "v"(b), [c] "v"(c), [q] "v"(q), [r] "v"(r), [v] "v"(v) : "vcc", "s20", "s21");')
    lines.append("  }")
    lines.append("  if (" + " && ".join(test.format(i=i) for i in range(8)) + ") out[0] = 1.f;")
    lines.append("}")
    return "\n".join(lines) + "\n"


def main():
    w = sys.stdout.write
    w(HEAD)
    for n, (_, kind, ins) in enumerate(PROBES):
        w(kernel(n, kind, ins))
    w("\ntypedef void (*kern_t)(float*);\nstruct Entry { const char* name; kern_t fn; };\n\nint main() {\n")
    w("  Entry es[] = {" + ", ".join(f'{{"{name}", probe_{n}}}' for n, (name, _, _) in enumerate(PROBES)) + "};\n")
    w(TAIL)


if __name__ == "__main__":
    main()
