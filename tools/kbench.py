#!/usr/bin/env python3
"""Per-kernel-family timing of one HRNet forward (library hipEvent profiler).  Usage: python tools/kbench.py [prec] [B] [V] [S]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "highres-net_amd"))
import torch
import bench
from hrnet_hip import binding
from DeepNetworks.HRNet import HRNet
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, V, S = (int(x) for x in (sys.argv[2:5] + ["32", "32", "128"][len(sys.argv[2:5]):]))
torch.manual_seed(1234)
net = HRNet(dict(bench.NETWORK, precision=prec)).cuda().eval()
lrs, alphas = bench.synth_inputs(B, V, S, "cuda", 100)
packed, dt = net.packed_parameters()
sr = torch.empty((B, 1, 3 * S, 3 * S), device="cuda")
for _ in range(2):
    binding.hrnet_forward(packed, dt, 2, True, lrs, alphas, out=sr)
torch.cuda.synchronize()
binding.profile_enable(True)
n = 3
for _ in range(n):
    binding.hrnet_forward(packed, dt, 2, True, lrs, alphas, out=sr)
torch.cuda.synchronize()
binding.profile_enable(False)
prof = binding.profile_read()
tot = sum(v["ms"] for v in prof.values()) / n
print(f"[{os.environ.get('HRN_CONV_DBG','-')}] {prec} B={B} V={V} S={S}: {tot:.3f} ms/fwd  {B / tot * 1e3:.1f} frames/s")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"   {k:24s} {v['launches'] // n:3d} x {v['ms'] / v['launches']:8.4f} ms  {v['flops'] / v['ms'] / 1e9:8.1f} TF/s  {v['bytes'] / v['ms'] / 1e6:8.1f} GB/s")
