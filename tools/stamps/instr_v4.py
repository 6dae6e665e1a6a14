#!/usr/bin/env python3
"""instr_v4.py <dir>: add s_memtime stamps to <dir>/conv3x3_v4.hip (diagnostic build for scratch/stamps_v4.py)"""
import sys
fn = sys.argv[1] + "/conv3x3_v4.hip"
s = open(fn).read()
s = s.replace("__device__ __attribute__((aligned(16))) unsigned hrn_v4_zero16[4];",
"""__device__ __attribute__((aligned(16))) unsigned hrn_v4_zero16[4];
__device__ unsigned long long hrn_v4_stamps[256 * 8 * 16];
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tl == 3 && c == 1) { if (tg == 0) st[i] = t_; else if (tg == 1) st[7 + (i)] = t_; } } while (0)""")
s = s.replace("    f32x16 acc[NCB][2];                                     // [cout block][pixel row]",
"    unsigned long long st[14];\n#pragma unroll\n    for (int i = 0; i < 14; ++i) st[i] = 0;\n    f32x16 acc[NCB][2];                                     // [cout block][pixel row]")
s = s.replace("                int issued = n_w;\n", "                STAMP(0);\n                int issued = n_w;\n")
s = s.replace("                // Hand-issued fragment reads (see conv3x3_r64.hip", "                STAMP(1);\n                // Hand-issued fragment reads (see conv3x3_r64.hip")
s = s.replace("                            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][cb], fb[s_][pb], acc[cb][pb], 0, 0, 0);",
"                            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][cb], fb[s_][pb], acc[cb][pb], 0, 0, 0);\n                            if (i == 0 && g == 0) STAMP(2);")
s = s.replace("                if (w < 4) stage_issue();\n", "                STAMP(3);\n                if (w < 4) stage_issue();\n                STAMP(6);\n")
s = s.replace("                wait_vm(issued);\n", "                wait_vm(issued);\n                STAMP(4);\n")
s = s.replace("                lds_done_then_barrier4();\n            };\n            stage(", "                lds_done_then_barrier4();\n                STAMP(5);\n            };\n            stage(")
s = s.replace("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");       // the last stages' look-ahead DMAs must not outlive the workgroup",
"    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");       // the last stages' look-ahead DMAs must not outlive the workgroup\n    if (COUT == 128 && RESM == 2 && lane == 0) {\n#pragma unroll\n        for (int i = 0; i < 14; ++i) hrn_v4_stamps[(bid * 8 + w) * 16 + i] = st[i];\n    }")
s = s.replace("}  // namespace\n\n// bf16, 128 input channels.", "}  // namespace\nextern \"C\" int hrn_dbg_read_stamps_v4(void* dst, size_t bytes) {\n    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrn_v4_stamps), bytes, 0, hipMemcpyDeviceToHost);\n}\n\n// bf16, 128 input channels.")
assert s.count("STAMP(") == 8, s.count("STAMP(")
open(fn, "w").write(s)
