#!/usr/bin/env python3
"""Save the bf16 forward's output of the library in use (HRNET_HIP_LIB) to gpurun_out/sr_NAME.pt, or compare two such files.
Usage: python tools/ab_out.py save NAME | python tools/ab_out.py cmp A B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "highres-net_amd"))
import torch
if sys.argv[1] == "save":
    import bench
    from DeepNetworks.HRNet import HRNet
    torch.manual_seed(1234)
    net = HRNet(dict(bench.NETWORK, precision="bf16")).cuda().eval()
    lrs, alphas = bench.synth_inputs(8, 9, 128, "cuda", 100)
    alphas[:, 7:] = 0.5
    with torch.no_grad():
        sr = net(lrs, alphas)
    torch.save(sr.cpu(), os.path.join(ROOT, "gpurun_out", f"sr_{sys.argv[2]}.pt"))
else:
    a, b = (torch.load(os.path.join(ROOT, "gpurun_out", f"sr_{n}.pt")) for n in sys.argv[2:4])
    d = (a - b).abs().max().item()
    print(f"{sys.argv[2]} vs {sys.argv[3]}: max |diff| {d:.3e} of max |a| {a.abs().max().item():.3e}; identical: {bool((a == b).all())}")
