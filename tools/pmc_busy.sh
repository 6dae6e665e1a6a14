#!/bin/bash
# rocprofv3 --pmc pass of tools/kbench.py bf16: effective clock and matrix-pipe busy fraction per kernel
# (profiles/r01_final_pmc_mfma_busy_clock.txt); on the GPU box: gpurun -- tools/pmc_busy.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_busy; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p -o t -- python3 $R/tools/kbench.py bf16 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
for f in $(find $O/p -name "*counter_collection.csv"); do cp $f $O/cc.csv; done
for f in $(find $O/p -name "*kernel_trace.csv"); do cp $f $O/kt.csv; done
rm -rf $O/p
cd $R
python3 - <<'PY'
import csv, collections, os, re
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_busy"
dur = {}
for r in csv.DictReader(open(O + "/kt.csv")):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(O + "/cc.csv")):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:50]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[k]["ns"] += dur.get(r["Dispatch_Id"], (0, ""))[0]; acc[k]["n"] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["ns"])[:10]
for k, c in rows:
    clk = c["GRBM_GUI_ACTIVE"] / 8 / max(c["ns"], 1)     # GHz
    # MFMA busy cycles summed over 1024 SIMDs; wall cycles = ns * clk
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / max(c["ns"] * clk, 1)
    print(f"{k:52s} n={int(c['n']):4d} total={c['ns']/1e6:8.2f} ms clk={clk:5.2f} GHz  mfma_busy={util:5.2f}")
PY
