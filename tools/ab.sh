#!/bin/bash
# usage (on the GPU box): tools/ab.sh VARIANT...  - A/B timing of kernel variants in ONE gpurun call (boxes differ by 3-5 %):
# tools/kbench.py for each variant, twice, round-robin; "prod" = the in-tree library, any other name = scratch/x/NAME/lib.so
# (tools/v6_abl.sh builds such libraries).  Prints ms per forward and the average launch of every conv family.
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = prod ]; then unset HRNET_HIP_LIB; else export HRNET_HIP_LIB=scratch/x/$v/lib.so; fi
  python tools/kbench.py bf16 > gpurun_out/ab_$v.txt 2>&1 || { echo "FAIL $v"; tail -5 gpurun_out/ab_$v.txt; exit 1; }
  f() { grep -E "$1" gpurun_out/ab_$v.txt | awk '{print $4}'; }
  echo "$v: $(grep 'ms/fwd' gpurun_out/ab_$v.txt | sed 's/.*: //') | 128x128+res $(f '128x128\+res') | 128x128 $(f '128x128 ') | 128x64+res $(f '128x64\+res') | 64x64 $(f '64x64 ') | 64x64+res $(f '64x64\+res')"
done; done
