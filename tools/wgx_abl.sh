#!/bin/bash
# Timing-only ablations of conv_wgrad_x3_kernel (wgrad_x3.hip, -DWGX_ABL=<bits>): builds scratch/x/wgx_<bits>/lib.so and times the
# weight-gradient launches of one bf16x3 train step (tools/train_prof.py).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
CS=highres-net_amd/hrnet_hip/csrc; B=highres-net_amd/hrnet_hip/build
for a in ${WGX_LIST:-0 1 2 4 8 3 7}; do
  d=scratch/x/wgx_$a; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWGX_ABL=$a -c $CS/wgrad_x3.hip -o $d/wgrad_x3.o
  objs=$(ls $B/*.o | grep -v wgrad_x3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/lib.so $objs $d/wgrad_x3.o
  echo "WGX_ABL=$a: $(HRNET_HIP_LIB=$PWD/$d/lib.so python tools/train_prof.py bf16x3 2>/dev/null | grep conv_wgrad_bf16x3)"
done
