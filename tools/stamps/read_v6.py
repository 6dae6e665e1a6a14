#!/usr/bin/env python3
"""Read the conv3x3_v6 stamps of a diagnostic build (-DV6_STAMP; HRNET_HIP_LIB=scratch/x/v6_stamp/lib.so): tile 1 of the largest
128 -> 128 + residual launch; per stage of chunk 1: MFMA loop, counted DMA wait, barrier; the tile's last stage and epilogue."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "highres-net_amd"))
import numpy as np, torch
import bench
from hrnet_hip import binding
from DeepNetworks.HRNet import HRNet
net = HRNet(dict(bench.NETWORK, precision="bf16")).cuda().eval()
lrs, alphas = bench.synth_inputs(32, 32, 128, "cuda", 100)
packed, dt = net.packed_parameters()
sr = torch.empty((32, 1, 384, 384), device="cuda")
lib = ctypes.CDLL(binding.LIB_PATH)
for _ in range(4):
    binding.hrnet_forward(packed, dt, 2, True, lrs, alphas, out=sr)
torch.cuda.synchronize()
buf = np.zeros((256, 8, 24), dtype=np.uint64)
assert lib.hrn_dbg_read_stamps_v6(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
s = buf.astype(np.int64)
s = s[s[:, 0, 20] > 0]
print("workgroups with stamps:", len(s))
def med(x): return float(np.median(x))
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    g = s[:, sl, :]
    for tg in range(3):
        o = 4 * tg
        print(f" {grp} chunk 1 tg={tg}: mfma_loop={med(g[..., o+1]-g[..., o+0]):6.0f} wait_vm={med(g[..., o+2]-g[..., o+1]):6.0f} barrier={med(g[..., o+3]-g[..., o+2]):6.0f} total={med(g[..., o+3]-g[..., o+0]):6.0f}")
    print(f" {grp} last stage : mfma_loop={med(g[..., 13]-g[..., 12]):6.0f} wait_vm={med(g[..., 14]-g[..., 13]):6.0f} epilogue={med(g[..., 16]-g[..., 14]):6.0f} barrier={med(g[..., 15]-g[..., 16]):6.0f} total={med(g[..., 15]-g[..., 12]):6.0f}")
print(f" tile 1 start -> tile 2 start: {med(s[..., 21]-s[..., 20]):7.0f}")
