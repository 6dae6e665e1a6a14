"""Per-kernel-family time of one train step (bench.py's step) under the library's hipEvent profiler: python tools/train_prof.py [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "highres-net_amd"), ROOT]
import torch
import bench
from hrnet_hip import binding
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
step = bench.make_train_step(dev, 32, 32, 64, precision=prec)
for _ in range(2):
    step()
torch.cuda.synchronize()
pf = bench.profile_families(binding, dev, step, 1)
tot = sum(v["ms"] for v in pf.values())
print(f"{prec}: profiled families {tot:.1f} ms (the step also runs unprofiled elementwise kernels)")
for k, v in sorted(pf.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:34s} {v['launches']:4d} x {v['ms'] / v['launches'] * 1e3:9.1f} us = {v['ms']:7.2f} ms   {v['flops'] / max(v['ms'], 1e-9) / 1e9:8.1f} TFLOP/s {v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s")
