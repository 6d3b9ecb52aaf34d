#!/usr/bin/env python3
"""Micro-benchmark generator (round 6): does the float64 add stream of the shift-reuse loop lose issue slots to
VGPR BANK CONFLICTS between the accumulator pair and the window operand pair?  The loop's operand register is
WIN + idx + 2k with a data-dependent even idx, the accumulator ACC + 12 g + 2 k: half of the adds read both
pairs from the same two banks (register number mod 4).

Bodies of 48 independent v_add_f64 acc, win, acc at two wavefronts per SIMD:
  same     win = acc's own register (what tools/micro/issue_cost.hip measures)
  bank0    win register = acc register mod 4 (same banks)
  bank2    win register = acc register + 2 mod 4 (the other two banks)
  idx0/2   the same through VGPR-index mode (s_set_gpr_idx_on every six adds), index 0 / 2
usage: python gen_add_banks.py > /tmp/add_banks.hip && hipcc --offload-arch=gfx950 -O3 -o add_banks /tmp/add_banks.hip"""

ACC, WIN = 40, 144          # 48 accumulator pairs v40..v135, window v144..v191


def body(mode):
    lines = []
    for i in range(48):
        a = ACC + 2 * i
        if mode == "same":
            w = a
        else:
            w = WIN + (2 * i) % 40
            if (w - a) % 4 != (0 if mode in ("bank0", "idx0", "idx2") else 2):
                w += 2
        if mode in ("idx0", "idx2") and i % 6 == 0:
            lines.append(f"s_mov_b32 s40, {0 if mode == 'idx0' else 2}")
            lines.append("s_set_gpr_idx_on s40, 1")
        lines.append(f"v_add_f64 v[{a}:{a + 1}], v[{w}:{w + 1}], v[{a}:{a + 1}]")
    return "\\n\\t".join(lines) + "\\n\\t"


MODES = ["same", "bank0", "bank2", "idx0", "idx2"]
print("#include <hip/hip_runtime.h>\n#include <cstdio>")
print('#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e_)); return 1; } } while (0)')
clob = ", ".join(f'"v{r}"' for r in range(40, 192)) + ', "s40", "vcc", "scc", "m0"'
print("template <int MODE> __global__ __launch_bounds__(256) void k(int iters, double *out) {\n    int n = iters;")
for m, name in enumerate(MODES):
    print(f'    if (MODE == {m}) asm volatile("Lk_%=:\\n\\t{body(name)}s_sub_u32 %0, %0, 1\\n\\ts_cmp_lg_u32 %0, 0\\n\\t'
          f's_cbranch_scc1 Lk_%=\\n\\ts_set_gpr_idx_off" : "+s"(n) : : {clob});')
print("    if (threadIdx.x == 1000) out[0] = (double)n;\n}")
print("int main() {\n    double *out; CK(hipMalloc(&out, 64));\n    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));")
print("    const int iters = 20000;\n    for (int wps = 1; wps <= 2; ++wps)")
print("        for (int mode = 0; mode < %d; ++mode) {\n            float best = 1e30f;" % len(MODES))
print("            for (int rep = 0; rep < 3; ++rep) {\n                CK(hipEventRecord(e0));\n                switch (mode) {")
for m in range(len(MODES)):
    print(f"                    case {m}: hipLaunchKernelGGL(k<{m}>, dim3(256 * wps), dim3(256), 0, 0, iters, out); break;")
print("                }\n                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));\n                float ms; CK(hipEventElapsedTime(&ms, e0, e1));")
print("                if (rep) best = ms < best ? ms : best;\n            }")
names = ", ".join(f'"{n}"' for n in MODES)
print(f"            const char *names[] = {{{names}}};")
print('            printf("%d wave(s)/SIMD %-6s %8.1f ns per 48 adds per wave, per SIMD %7.1f ns = %6.1f clk at 2.4 GHz (ideal 192)\\n", wps, names[mode], best * 1e6 / iters, best * 1e6 / iters / wps, best * 1e6 / iters / wps * 2.4);')
print("        }\n    return 0;\n}")
