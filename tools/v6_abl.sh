#!/bin/bash
# tools/v6_abl.sh BITS...: timing-only ablation builds of conv3x3_v6.hip (-DV6_ABL=BITS, see the kernel's header) linked with the
# in-tree objects -> scratch/x/v6_BITS/lib.so; on the GPU box: HRNET_HIP_LIB=scratch/x/v6_BITS/lib.so python tools/kbench.py bf16
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/highres-net_amd/hrnet_hip/build
for bits in "$@"; do
  D=$ROOT/scratch/x/v6_$bits; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -DV6_ABL=$bits -c $ROOT/highres-net_amd/hrnet_hip/csrc/conv3x3_v6.hip -o $D/conv3x3_v6.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib.so $(ls $B/*.o | grep -v conv3x3_v6.o) $D/conv3x3_v6.o
  echo $D/lib.so
done
