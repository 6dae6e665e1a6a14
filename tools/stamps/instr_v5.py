#!/usr/bin/env python3
"""instr_v5.py <dir>: add s_memtime stamps to <dir>/conv3x3_v5.hip (a COPY of csrc/; diagnostic build for read_v5.py).
Stage stamps (tile 3, chunk 1, tap rows 0 and 1): 0 stage start, 1 early DMA issue done, 2 first MFMA issued, 3 MFMAs done,
6 late DMA issue done, 4 counted vmcnt passed, 5 barrier passed.  Tile stamps (tile 2): start, epilogue start, epilogue end, next tile."""
import sys
fn = sys.argv[1] + "/conv3x3_v5.hip"
s = open(fn).read()
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
rep("__device__ __attribute__((aligned(16))) unsigned hrn_v5_zero16[4];",
    """__device__ __attribute__((aligned(16))) unsigned hrn_v5_zero16[4];
__device__ unsigned long long hrn_v5_stamps[256 * 8 * 20];
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tl == 3 && c == 1) { if (tg == 0) st[i] = t_; else if (tg == 1) st[7 + (i)] = t_; } } while (0)""")
rep("    f32x4 acc[8][4];", "    unsigned long long st[14], tt[4] = {0, 0, 0, 0};\n#pragma unroll\n    for (int i = 0; i < 14; ++i) st[i] = 0;\n    f32x4 acc[8][4];")
rep("        const bool more_tiles = tl + 1 < ntl;", "        if (tl == 2) tt[0] = __builtin_amdgcn_s_memtime();\n        if (tl == 3) tt[3] = __builtin_amdgcn_s_memtime();\n        const bool more_tiles = tl + 1 < ntl;")
rep("                int issued = 3;\n", "                STAMP(0);\n                int issued = 3;\n")
rep("                if (w >= 4) stage_issue();\n", "                if (w >= 4) stage_issue();\n                STAMP(1);\n")
rep("                        acc[qt * 2 + k][pxb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][k], fb[bs][pxb], acc[qt * 2 + k][pxb], 0, 0, 0);",
    "                        acc[qt * 2 + k][pxb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][k], fb[bs][pxb], acc[qt * 2 + k][pxb], 0, 0, 0);\n                        if (i == 0 && g == 0) STAMP(2);")
rep("                if (w < 4) stage_issue();\n", "                STAMP(3);\n                if (w < 4) stage_issue();\n                STAMP(6);\n")
rep("                wait_vm5(issued);\n", "                wait_vm5(issued);\n                STAMP(4);\n")
rep("                if (c == 3 && tg == 2) {\n", "                if (c == 3 && tg == 2) {\n                    if (tl == 2) tt[1] = __builtin_amdgcn_s_memtime();\n")
rep("                    else epilogue(std::integral_constant<int, 2>{});\n                }", "                    else epilogue(std::integral_constant<int, 2>{});\n                    if (tl == 2) tt[2] = __builtin_amdgcn_s_memtime();\n                }")
rep("                barrier5();\n            };", "                barrier5();\n                STAMP(5);\n            };")
rep("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n}", """    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!RES && lane == 0) {
#pragma unroll
        for (int i = 0; i < 14; ++i) hrn_v5_stamps[(bid * 8 + w) * 20 + i] = st[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) hrn_v5_stamps[(bid * 8 + w) * 20 + 14 + i] = tt[i];
    }
}""")
rep("}  // namespace\n\n// bf16 128 -> 128, residual none", """}  // namespace
extern "C" int hrn_dbg_read_stamps_v5(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrn_v5_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

// bf16 128 -> 128, residual none""")
open(fn, "w").write(s)
