"""Determinism soak: repeated forwards (bf16, and bf16x3 with `python tools/soak.py bf16x3`) at the bench size and two ragged sizes
must be bit-identical (the conv kernels wait on hand-counted lgkmcnt / vmcnt values; a wrong count would show up as a
timing-dependent mismatch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "highres-net_amd"))
import torch, bench
from DeepNetworks.HRNet import HRNet
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
net = HRNet(dict(bench.NETWORK, precision=prec)).cuda().eval()
bad = 0
for B, V, S in ((32, 32, 128), (5, 9, 100), (3, 7, 50)):
    lrs, alphas = bench.synth_inputs(B, V, S, "cuda", 7)
    with torch.no_grad():
        ref = net(lrs, alphas).clone()
        for i in range((60 if S == 128 else 200) // (3 if prec == "bf16x3" else 1)):
            y = net(lrs, alphas)
            if not torch.equal(y, ref):
                bad += 1
                print("MISMATCH", B, V, S, i, float((y - ref).abs().max()))
    print(B, V, S, "done")
print("soak mismatches:", bad)
