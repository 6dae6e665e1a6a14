#!/bin/bash
# A/B of fc1_mfma_kernel's decomposition (slices, neuron blocks per wave, loads in flight): builds scratch/x/fc1_<cfg>/lib.so from the
# tree's objects + a shiftnet.o compiled with the variant's macros, then times ShiftNet's fc1 on one GPU (run on the GPU box).
set -e
cd "$(dirname "$0")/.."
CS=highres-net_amd/hrnet_hip/csrc; B=highres-net_amd/hrnet_hip/build
for cfg in "32 1 2" "64 1 2" "64 2 2" "32 2 2" "64 2 4" "128 2 2" "64 4 2" "32 1 4"; do
  set -- $cfg; d=scratch/x/fc1_$1_$2_$3; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DFC_SPLIT_N=$1 -DFC_NBW=$2 -DFC_DEPTH=$3 -c $CS/shiftnet.hip -o $d/shiftnet.o
  objs=$(ls $B/*.o | grep -v shiftnet.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/lib.so $objs $d/shiftnet.o
  echo "cfg split=$1 nbw=$2 depth=$3: $(HRNET_HIP_LIB=$PWD/$d/lib.so python tools/fc1_time.py)"
done
