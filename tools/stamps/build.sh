#!/bin/bash
# build.sh <name> r64: copy csrc/ to scratch/x/<name>, instrument one kernel, link scratch/x/<name>/lib.so
# then on the GPU:  HRNET_HIP_LIB=scratch/x/<name>/lib.so python tools/stamps/read_r64.py   (conv3x3_v6: -DV6_STAMP + tools/stamps/read_v6.py)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
D=$ROOT/scratch/x/$1
rm -rf $D && mkdir -p $D
cp $ROOT/highres-net_amd/hrnet_hip/csrc/*.h $ROOT/highres-net_amd/hrnet_hip/csrc/*.hip $D/
python3 $ROOT/tools/stamps/instr_$2.py $D
cd $D
ls *.hip | xargs -P 8 -I{} /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -c {} -o {}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib.so *.hip.o
echo $D/lib.so
