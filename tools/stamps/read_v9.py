#!/usr/bin/env python3
"""Read the conv3x3_v9 stamps of a diagnostic build (-DV9_STAMP; HRNET_HIP_LIB=scratch/x/v9_stamp/lib.so, HRN_CONV_V9=1): phases 4 and 5
of the largest 128 -> 128 + residual launch.  Per segment and role: work (MFMA stream | DMA issue + epilogue), counted DMA wait, time
at the barrier; the phase length."""
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
buf = np.zeros((256, 8, 80), dtype=np.uint64)
assert lib.hrn_dbg_read_stamps_v9(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
s = buf.astype(np.int64)
s = s[s[:, 0, 0] > 0]
print("workgroups with stamps:", len(s))
def med(x): return float(np.median(x))
# phase 4: team 0 (waves 0-3) ON, team 1 OFF; phase 5: the other way round.  stamps: 36*(ph&1) + 3*seg + {0 start, 1 work done (OFF only), 2 at barrier}
for ph in (4, 5):
    b = 36 * (ph & 1)
    on = slice(0, 4) if ph % 2 == 0 else slice(4, 8)
    off = slice(4, 8) if ph % 2 == 0 else slice(0, 4)
    print(f"phase {ph}:")
    for seg in range(12):
        o = b + 3 * seg
        nxt = b + 3 * (seg + 1) if seg < 11 else None
        on_work = med(s[:, on, o + 2] - s[:, on, o])
        off_work = med(s[:, off, o + 1] - s[:, off, o])
        off_wait = med(s[:, off, o + 2] - s[:, off, o + 1])
        line = f"  seg {seg:2d} (c={seg // 3} tg={seg % 3}): ON mfma stream {on_work:6.0f} | OFF issue+epilogue {off_work:6.0f} dma wait {off_wait:6.0f}"
        if nxt is not None:
            line += f" | segment length ON {med(s[:, on, nxt] - s[:, on, o]):6.0f} OFF {med(s[:, off, nxt] - s[:, off, o]):6.0f}"
        print(line)
    print(f"  phase length (seg 0 start -> seg 11 barrier): ON {med(s[:, on, b + 35] - s[:, on, b]):7.0f}")
