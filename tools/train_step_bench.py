#!/usr/bin/env python3
"""Time one optimisation step of src/train.py:164-191 on the HIP modules:

    srs = fusion_model(lrs, alphas); shifts = register_batch(regis_model, crops, reference); srs_shifted = apply_shifts(...);
    loss = -cPSNR(srs_shifted, hrs, mask) + lambda mean(shifts)^2; loss.backward(); optimizer.step()

at the reference's training shape (config/config.json: batch 32, up to 32 views, 64 x 64 patches) with synthetic data.
usage: python tools/train_step_bench.py [B V S steps] [--torch-adam]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "highres-net_amd"))
import numpy as np
import torch

from oracle import synth, weights            # seeded weights / synthetic inputs only (no oracle arithmetic on the path)
from DeepNetworks.HRNet import HRNet
from DeepNetworks.ShiftNet import ShiftNet
from hrnet_hip.optim import FusedAdam


def register_batch(shiftNet, lrs, reference):                 # train.py:26-44
    return torch.stack([shiftNet(torch.cat([reference, lrs[:, i:i + 1]], 1)) for i in range(lrs.size(1))], 1)


def apply_shifts(shiftNet, images, thetas, device):           # train.py:47-63
    b, n, h, w = images.shape
    return shiftNet.transform(thetas.view(-1, 2), images.view(-1, 1, h, w), device=device).view(-1, n, h, w)


def get_loss_cpsnr(srs, hrs, hr_maps):                        # train.py:66-87
    nclear = torch.sum(hr_maps, dim=(1, 2))
    bright = torch.sum(hr_maps * (hrs - srs), dim=(1, 2)).clone().detach() / nclear
    return -10 * torch.log10(torch.sum(hr_maps * (srs + bright.view(-1, 1, 1) - hrs) ** 2, dim=(1, 2)) / nclear)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, V, S, steps = (int(a) for a in args[:4]) if len(args) >= 4 else (32, 32, 64, 5)
    dev = torch.device("cuda:0")
    lrs, alphas = synth.fast_batch(3, B, V, S)
    rng = np.random.Generator(np.random.PCG64(1))
    hrs = torch.from_numpy((rng.random((B, 3 * S, 3 * S), dtype=np.float32) * 0.25)).to(dev)
    maps = torch.ones((B, 3 * S, 3 * S), device=dev)
    maps[:, :3] = 0; maps[:, -3:] = 0; maps[:, :, :3] = 0; maps[:, :, -3:] = 0
    x, a = torch.from_numpy(lrs).to(dev), torch.from_numpy(alphas).to(dev)
    fusion = HRNet({k: dict(v) for k, v in weights.HRNET_CONFIG.items()})
    fusion.load_state_dict(weights.to_torch_state(weights.hrnet_state(1234)))
    regis = ShiftNet()
    regis.load_state_dict(weights.to_torch_state(weights.shiftnet_state(4321)))
    fusion, regis = fusion.to(dev).train(), regis.to(dev).train()
    params = list(fusion.parameters()) + list(regis.parameters())
    opt = torch.optim.Adam(params, lr=1e-4) if "--torch-adam" in sys.argv else FusedAdam(params, lr=1e-4)
    off = (3 * S - 128) // 2

    def step():
        opt.zero_grad()
        srs = fusion(x, a)
        shifts = register_batch(regis, srs[:, :, off:off + 128, off:off + 128], hrs[:, off:off + 128, off:off + 128].reshape(-1, 1, 128, 128))
        shifted = apply_shifts(regis, srs, shifts, dev)[:, 0]
        loss = -get_loss_cpsnr(shifted, hrs, maps)
        loss = torch.mean(loss) + 1e-6 * torch.mean(shifts) ** 2
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    print(f"train step B={B} V={V} S={S}: {dt * 1e3:.1f} ms/step ({B / dt:.0f} samples/s), loss {float(loss.detach()):.3f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, optimiser {type(opt).__name__}")


if __name__ == "__main__":
    main()
