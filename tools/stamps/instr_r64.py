#!/usr/bin/env python3
"""instr_r64.py <dir>: add s_memtime stamps to <dir>/conv3x3_r64.hip (a COPY of csrc/, diagnostic build for read_r64.py).

Stamps per wave for phases 40 and 41 of the persistent loop: 0 phase start, 1 work done, 2 barrier passed; ON phase: 6 K loop
starts; OFF phase: 3 DMAs issued, 4 epilogue done, 5 DMAs landed.  They live in SGPRs and are written once at kernel end, so
no store disturbs the counted vmcnt waits.  hrn_dbg_set_abl(bits): 1 skip the output stores, 2 skip the halo DMAs (timing only).
"""
import sys
fn = sys.argv[1] + "/conv3x3_r64.hip"
s = open(fn).read()
def rep(old, new, count=1):
    global s
    assert old in s, old
    s = s.replace(old, new, count)
rep("__device__ __attribute__((aligned(16))) unsigned hrn_r64_zero16[4];",
    """__device__ __attribute__((aligned(16))) unsigned hrn_r64_zero16[4];
__device__ unsigned long long hrn_r64_stamps[2][256 * 8 * 16];
__device__ int hrn_r64_abl;
constexpr int PH0 = 40;
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (cur_ph == PH0) st[i] = t_; else if (cur_ph == PH0 + 1) st[7 + (i)] = t_; } while (0)""")
rep("    const bool has_slope = p.slope != nullptr;",
    "    unsigned long long st[14];\n#pragma unroll\n    for (int i = 0; i < 14; ++i) st[i] = 0;\n    int cur_ph = -1;\n    const int abl = hrn_r64_abl;\n    const bool has_slope = p.slope != nullptr;")
rep("        constexpr int NK = 36;", "        STAMP(6);\n        constexpr int NK = 36;")
rep("        if (more) issue(cur_m, cur_t);                      // in flight while the epilogue runs", "        if (more && !(abl & 2)) issue(cur_m, cur_t);\n        STAMP(3);")
rep("        {   // the halo DMAs were issued before this tile's stores", "        STAMP(4);\n        {   // the halo DMAs were issued before this tile's stores")
rep("        fix_borders();\n    };", "        fix_borders();\n        STAMP(5);\n    };")
rep("    for (int ph = 0; ph <= ntl; ++ph) {\n        const int q = ph - team;", "    for (int ph = 0; ph <= ntl; ++ph) {\n        cur_ph = ph;\n        STAMP(0);\n        const int q = ph - team;")
rep("        if (ph < ntl) lds_done_then_barrier();\n    }\n}", """        STAMP(1);
        if (ph < ntl) lds_done_then_barrier();
        STAMP(2);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 14; ++i) hrn_r64_stamps[RES ? 1 : 0][(bid * 8 + wave) * 16 + i] = st[i];
    }
}""")
s = s.replace("if (gy < H && gx < W) op[g] = u;", "if (gy < H && gx < W && !(abl & 1)) op[g] = u;")
s = s.replace("if (gy < H && x0 + (r & ~3) + j < W) *(u32x4*)(oq + j * 128) = uu[j];", "if (gy < H && x0 + (r & ~3) + j < W && !(abl & 1)) *(u32x4*)(oq + j * 128) = uu[j];")
rep("}  // namespace\n\n// bf16 64 -> 64 with a plain input", """}  // namespace
extern "C" int hrn_dbg_set_abl(int v) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(hrn_r64_abl), &v, sizeof(int), 0, hipMemcpyHostToDevice);
}
extern "C" int hrn_dbg_read_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrn_r64_stamps), bytes, 0, hipMemcpyDeviceToHost);
}

// bf16 64 -> 64 with a plain input""")
open(fn, "w").write(s)
