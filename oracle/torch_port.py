"""Own PyTorch-CPU port of HRNet.forward, used ONLY as bench.py's `cpu_baseline` (kind "port") and checked
against the goldens in tests/test_oracle_golden.py.  TEST / MEASUREMENT INFRASTRUCTURE, never the product path.

Why it exists: the north star asks for "the reference's CPU PyTorch path timed on the host cores of the same
box".  The reference's Python cannot travel to the GPU box, so this restatement issues the same ATen/oneDNN
operator sequence (conv2d, prelu, conv_transpose2d, median) from a state dict, written from the algorithm
description (HRNet.py:186-211), and is timed there with a stated thread count.
"""
import torch
import torch.nn.functional as F


def _res_block(x, st, pre):
    t = F.prelu(F.conv2d(x, st[pre + ".block.0.weight"], st[pre + ".block.0.bias"], padding=1), st[pre + ".block.1.weight"])
    t = F.prelu(F.conv2d(t, st[pre + ".block.2.weight"], st[pre + ".block.2.bias"], padding=1), st[pre + ".block.3.weight"])
    return x + t


@torch.no_grad()
def hrnet_forward(lrs, alphas, st, num_layers=2, alpha_residual=True):
    """lrs (B,V,H,W), alphas (B,V) float32 CPU tensors; st: state dict of float32 CPU tensors -> (B,1,3H,3W)."""
    b, v, h, w = lrs.shape
    ref = torch.median(lrs[:, :9], 1, keepdim=True).values                      # lower median, pads included
    x = torch.stack([lrs, ref.expand(-1, v, -1, -1)], 2).reshape(b * v, 2, h, w)
    x = F.prelu(F.conv2d(x, st["encode.init_layer.0.weight"], st["encode.init_layer.0.bias"], padding=1), st["encode.init_layer.1.weight"])
    for i in range(num_layers):
        x = _res_block(x, st, f"encode.res_layers.{i}")
    x = F.conv2d(x, st["encode.final.0.weight"], st["encode.final.0.bias"], padding=1).reshape(b, v, 64, h, w)
    n = v
    while n // 2 > 0:
        parity, half = n % 2, n // 2
        alice = x[:, :half]
        bob = x[:, half:n - parity].flip(1)
        z = torch.cat([alice, bob], 2).reshape(b * half, 128, h, w)
        z = _res_block(z, st, "fuse.fuse.0")
        f = F.prelu(F.conv2d(z, st["fuse.fuse.1.weight"], st["fuse.fuse.1.bias"], padding=1), st["fuse.fuse.2.weight"])
        f = f.reshape(b, half, 64, h, w)
        if alpha_residual:
            a_bob = alphas[:, half:n - parity].flip(1).reshape(b, half, 1, 1, 1)
            f = alice + a_bob * f
        x, n = f, half
    x = x.mean(1)
    x = F.prelu(F.conv_transpose2d(x, st["decode.deconv.0.weight"], st["decode.deconv.0.bias"], stride=3), st["decode.deconv.1.weight"])
    return F.conv2d(x, st["decode.final.weight"], st["decode.final.bias"])
