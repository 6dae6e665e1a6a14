#!/usr/bin/env python3
"""tools/prof_collect.py TAG: turn gpurun_out/final/ (written by tools/prof_all.sh on the GPU box) into the committed evidence
profiles/TAG_*: bench lines, the rocprofv3 kernel stats, the HBM traffic summary + profiles/<bench.TRAFFIC_JSON> (bytes per launch
per kernel family, with a hash of the kernel sources so that bench.py never reports stale numbers), MFMA-busy / clock, stamps."""
import collections, csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
O, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
for src, dst in (("bench_steps20.json", "bench_steps20.json"), ("bench_train.json", "bench_train_steps5.json"), ("bench_steps5_kernel_stats.csv", "bench_steps5_kernel_stats.csv"),
                 ("bench_steps5_under_rocprof.json", "bench_steps5_under_rocprof.json"), ("kbench_c5.txt", "kbench_c5.txt"), ("v6_stamps.txt", "v6_stamps.txt"),
                 ("bench_x3_steps10.json", "bench_bf16x3_steps10.json"), ("bench_train_x3.json", "bench_train_bf16x3_steps5.json"),
                 ("x3_kernel_stats.csv", "bf16x3_forward_kernel_stats.csv"), ("train_kernel_stats.csv", "train_step_kernel_stats.csv"),
                 ("train_x3_kernel_stats.csv", "train_step_bf16x3_kernel_stats.csv"), ("kbench_x3.txt", "kbench_bf16x3.txt")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, f"{tag}_{dst}"))

FAMILY = [(r"conv3x3_v6_kernel<128, 128, 2, \w+, false>", "conv3x3_bf16_128x128+res"), (r"conv3x3_v6_kernel<128, 128, 0, \w+, false>", "conv3x3_bf16_128x128"),
          (r"conv3x3_v6_kernel<128, 64, 3, \w+, false>", "conv3x3_bf16_128x64+res"), (r"conv3x3_v6_kernel<128, 64, 0, \w+, false>", "conv3x3_bf16_128x64"),
          (r"conv3x3_r64_kernel<false>", "conv3x3_bf16_64x64"), (r"conv3x3_r64_kernel<true>", "conv3x3_bf16_64x64+res"),
          (r"stem_mfma_kernel", "stem2x64_bf16"), (r"decoder_kernel", "decoder_bf16")]


def family(name):
    for pat, fam in FAMILY:
        if re.search(pat, name):
            return fam, re.search(r"(\w+_kernel(<[^>]*>)?)", name).group(1)
    return None, None


def means(path, counter):
    acc = collections.defaultdict(list)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        fam, short = family(r["Kernel_Name"])
        if fam:
            acc[fam].append(float(r["Counter_Value"]))
            names[fam] = short
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}, names


fetch, names = means(os.path.join(O, "pmc_fetch_counter_collection.csv"), "FETCH_SIZE")
write, _ = means(os.path.join(O, "pmc_write_counter_collection.csv"), "WRITE_SIZE")
lines = ["# two separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/kbench.py bf16   and the same with WRITE_SIZE",
         "# (B=32, V=32, 128x128, bf16; 5 forwards each).  Per kernel: mean counter value per dispatch, in KB as rocprofv3 reports them.",
         "# bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE is doubled on gfx950 for 16-byte-per-lane reads (MI355X_MICROARCH.md)."]
per_launch = {}
for fam in fetch:
    f, n = fetch[fam]
    w = write.get(fam, (0.0, 0))[0]
    b = int((2 * f + w) * 1024)
    per_launch[fam] = b
    lines.append(f"{names[fam]:34s} -> {fam:26s} dispatches {n:3d}  FETCH_SIZE {f:12.1f}  WRITE_SIZE {w:12.1f}  bytes/launch {b:,}")
open(os.path.join(P, f"{tag}_hbm_traffic_pmc.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"_comment": "HBM bytes per average launch at the bench workload (B=32,V=32,128x128,bf16) from two separate rocprofv3 --pmc passes of "
                       "tools/kbench.py (tools/prof_all.sh), FETCH_SIZE doubled per MI355X_MICROARCH.md; raw means: profiles/" + tag + "_hbm_traffic_pmc.txt. "
                       "bench.py reports these as `traffic` only while kernel_source_hash matches the kernel sources in the tree.",
           "workload": {"batch": 32, "views": 32, "size": 128, "precision": "bf16"}, "kernel_source_hash": bench.kernel_source_hash(),
           "bytes_per_launch": per_launch}, open(os.path.join(P, bench.TRAFFIC_JSON), "w"), indent=2)

# MFMA busy / clock
dur = {}
for r in csv.DictReader(open(os.path.join(O, "pmc_busy_kt.csv"))):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(os.path.join(O, "pmc_busy_cc.csv"))):
    fam, short = family(r["Kernel_Name"])
    if not fam:
        continue
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[short]["ns"] += dur.get(r["Dispatch_Id"], 0)
        acc[short]["n"] += 1
out = ["# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -- python3 tools/kbench.py bf16",
       "# clk = GRBM_GUI_ACTIVE / 8 / duration (effective clock); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clk);",
       "# busy x clk = matrix-pipe cycles delivered per ns per SIMD: what the chip sustains at its power limit on this data"]
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
    clk = c["GRBM_GUI_ACTIVE"] / 8 / max(c["ns"], 1)
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / max(c["ns"] * clk, 1)
    out.append(f"{k:40s} n={int(c['n']):4d} total={c['ns'] / 1e6:8.2f} ms clk={clk:5.2f} GHz  mfma_busy={util:5.2f}  busy x clk={util * clk:5.2f}")
open(os.path.join(P, f"{tag}_pmc_mfma_busy_clock.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(lines[3:]))
print("\n".join(out[3:]))
