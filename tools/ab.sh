#!/bin/bash
# usage (GPU box): tools/ab.sh prod NAME...  - tools/kbench.py bf16 for the in-tree library (prod) and for scratch/x/NAME/lib.so
# (HRNET_HIP_LIB), twice, alternating, so that box-to-box and run-to-run noise does not decide an A/B; per-kernel lines in gpurun_out/ab_NAME.txt
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = prod ]; then unset HRNET_HIP_LIB; else export HRNET_HIP_LIB=scratch/x/$v/lib.so; fi
  python tools/kbench.py bf16 > gpurun_out/ab_$v.txt 2>&1 || { echo "FAIL $v"; tail -5 gpurun_out/ab_$v.txt; exit 1; }
  echo "$v: $(grep 'ms/fwd' gpurun_out/ab_$v.txt | sed 's/.*: //') | $(grep -E '64x64 ' gpurun_out/ab_$v.txt | awk '{print $5}' | tr '\n' ' ') | $(grep -E '64x64\+res' gpurun_out/ab_$v.txt | awk '{print $5}')"
done; done
