#!/usr/bin/env python3
"""Read the r64 phase stamps of the diagnostic build (HRNET_HIP_LIB=scratch/x/st/lib.so) after one bf16 forward."""
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
abl = int(os.environ.get("ABL", "0"))
if abl: assert lib.hrn_dbg_set_abl(abl) == 0
print("ABL", abl)
for _ in range(6):
    binding.hrnet_forward(packed, dt, 2, True, lrs, alphas, out=sr)
torch.cuda.synchronize()
buf = np.zeros((2, 256, 8, 16), dtype=np.uint64)
rc = lib.hrn_dbg_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
def med(x): return float(np.median(x))
for res in (0, 1):
    s = buf[res].astype(np.int64)                    # [wg][wave][16]
    print(f"=== RES={res}")
    for phase in (0, 1):
        o = 7 * phase
        for team in (0, 1):
            w = s[:, team * 4:(team + 1) * 4, :]
            q_even = ((40 + phase - team) & 1) == 0
            work = w[..., o + 1] - w[..., o + 0]
            barw = w[..., o + 2] - w[..., o + 1]
            line = f" ph={40+phase} team={team} {'ON ' if q_even else 'OFF'} work={med(work):7.0f} barrier_wait={med(barw):6.0f}"
            if q_even:
                line += f" bias_init={med(w[..., o + 6] - w[..., o + 0]):6.0f} kloop={med(w[..., o + 1] - w[..., o + 6]):7.0f}"
            else:
                line += (f" issue={med(w[..., o + 3] - w[..., o + 0]):6.0f} epilogue={med(w[..., o + 4] - w[..., o + 3]):6.0f}"
                         f" dma_wait={med(w[..., o + 5] - w[..., o + 4]):6.0f}")
            print(line)
    w = s[:, :, :]
    print(f" phase length (start ph41 - start ph40): {med(w[..., 7] - w[..., 0]):7.0f}   p10={np.percentile(w[...,7]-w[...,0],10):.0f} p90={np.percentile(w[...,7]-w[...,0],90):.0f}")
