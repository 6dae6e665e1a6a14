#!/usr/bin/env python3
"""Read the conv3x3_v5 stamps of a diagnostic build (tools/stamps/build.sh <name> v5; HRNET_HIP_LIB=scratch/x/<name>/lib.so)."""
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
buf = np.zeros((256, 8, 20), dtype=np.uint64)
assert lib.hrn_dbg_read_stamps_v5(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
s = buf.astype(np.int64)
s = s[s[:, 0, 0] > 0]
print("workgroups with stamps:", len(s))
def med(x): return float(np.median(x))
for grp, sl in (("waves 0-3 (MFMA first)", slice(0, 4)), ("waves 4-7 (DMA issue first)", slice(4, 8))):
    g = s[:, sl, :]
    for ph, name in ((0, "tg=0"), (1, "tg=1")):
        o = 7 * ph
        print(f" {grp} {name}: early_issue={med(g[..., o+1]-g[..., o+0]):6.0f} reads+wait={med(g[..., o+2]-g[..., o+1]):6.0f} "
              f"mfma={med(g[..., o+3]-g[..., o+2]):6.0f} late_issue={med(g[..., o+6]-g[..., o+3]):6.0f} wait_vm={med(g[..., o+4]-g[..., o+6]):6.0f} "
              f"barrier={med(g[..., o+5]-g[..., o+4]):6.0f} total={med(g[..., o+5]-g[..., o+0]):6.0f}")
t = buf.astype(np.int64)[:, :, 14:18]
t = t[(t[:, 0, 0] > 0) & (t[:, 0, 3] > 0)]
print("tiles with both stamps:", len(t))
print(f" tile start -> epilogue start: {med(t[..., 1]-t[..., 0]):7.0f}   epilogue: {med(t[..., 2]-t[..., 1]):7.0f}   epilogue end -> next tile start: {med(t[..., 3]-t[..., 2]):7.0f}   tile total: {med(t[..., 3]-t[..., 0]):7.0f}")
