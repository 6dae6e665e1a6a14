"""Own PyTorch-CPU port of HRNet.forward, used ONLY as bench.py's `cpu_baseline` (kind "port") and checked
against the goldens in tests/test_oracle_golden.py.  TEST / MEASUREMENT INFRASTRUCTURE, never the product path.

Why it exists: the north star asks for "the reference's CPU PyTorch path timed on the host cores of the same
box".  The reference's Python cannot travel to the GPU box, so this restatement issues the same ATen/oneDNN
operator sequence (conv2d, prelu, conv_transpose2d, median) from a state dict, written from the algorithm
description (HRNet.py:186-211), and is timed there with a stated thread count.
"""
import torch
import torch.nn.functional as F


# When a dict is installed here, every PReLU adds sum |g * min(x, 0)| of its backward pass under its slope's state-dict key: the
# single-slope gradient is the SIGNED sum of those terms, and sum |terms| is what bounds a float32 implementation's error on it
# (the gradient tests scale their scalar tolerances with it instead of with some other parameter's magnitude).
ABS_TERMS = None


def _prelu(x, st, key):
    y = F.prelu(x, st[key])
    if ABS_TERMS is not None and y.requires_grad:
        rec, neg = ABS_TERMS, x.detach().clamp(max=0)
        y.register_hook(lambda g: rec.__setitem__(key, rec.get(key, 0.0) + float((g * neg).abs().sum())))
    return y


def _res_block(x, st, pre):
    t = _prelu(F.conv2d(x, st[pre + ".block.0.weight"], st[pre + ".block.0.bias"], padding=1), st, pre + ".block.1.weight")
    t = _prelu(F.conv2d(t, st[pre + ".block.2.weight"], st[pre + ".block.2.bias"], padding=1), st, pre + ".block.3.weight")
    return x + t


@torch.no_grad()
def hrnet_forward(lrs, alphas, st, num_layers=2, alpha_residual=True):
    """lrs (B,V,H,W), alphas (B,V) float32 CPU tensors; st: state dict of float32 CPU tensors -> (B,1,3H,3W)."""
    b, v, h, w = lrs.shape
    ref = torch.median(lrs[:, :9], 1, keepdim=True).values                      # lower median, pads included
    x = torch.stack([lrs, ref.expand(-1, v, -1, -1)], 2).reshape(b * v, 2, h, w)
    x = _prelu(F.conv2d(x, st["encode.init_layer.0.weight"], st["encode.init_layer.0.bias"], padding=1), st, "encode.init_layer.1.weight")
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
        f = _prelu(F.conv2d(z, st["fuse.fuse.1.weight"], st["fuse.fuse.1.bias"], padding=1), st, "fuse.fuse.2.weight")
        f = f.reshape(b, half, 64, h, w)
        if alpha_residual:
            a_bob = alphas[:, half:n - parity].flip(1).reshape(b, half, 1, 1, 1)
            f = alice + a_bob * f
        x, n = f, half
    x = x.mean(1)
    x = _prelu(F.conv_transpose2d(x, st["decode.deconv.0.weight"], st["decode.deconv.0.bias"], stride=3), st, "decode.deconv.1.weight")
    y = F.conv2d(x, st["decode.final.weight"], st["decode.final.bias"])
    if ABS_TERMS is not None and y.requires_grad:
        rec = ABS_TERMS
        y.register_hook(lambda g: rec.__setitem__("decode.final.bias", rec.get("decode.final.bias", 0.0) + float(g.abs().sum())))
    return y


# ---------------------------------------------------------------------------------------------------------------------------
# Ports of the training-only pieces, used by the gradient tests as the torch-autograd oracle (fp64 on the CPU) and pinned to the
# reference's own train step by tests/golden/train_step.npz (tests/test_oracle_golden.py::test_train_step_port_matches_reference).


def shiftnet_forward_train(x, st, keep_mask):
    """ShiftNet.forward in train mode (ShiftNet.py:49-75: mean subtraction, 8 x conv + BatchNorm(batch statistics) + ReLU with
    pools after layers 2 / 4 / 6, dropout p = 0.5 with the GIVEN keep-mask (B, 32768), fc1 + ReLU, fc2)."""
    x = x - x.mean(dim=(2, 3), keepdim=True)
    for i in range(1, 9):
        x = F.conv2d(x, st[f"layer{i}.0.weight"], st[f"layer{i}.0.bias"], padding=1)
        x = F.batch_norm(x, None, None, st[f"layer{i}.1.weight"], st[f"layer{i}.1.bias"], training=True, eps=1e-5)
        x = F.relu(x)
        if i in (2, 4, 6):
            x = F.max_pool2d(x, 2)
    x = x.reshape(x.shape[0], -1) * keep_mask * 2.0
    x = F.relu(F.linear(x, st["fc1.weight"], st["fc1.bias"]))
    return F.linear(x, st["fc2.weight"])


def lanczos_shift(img, shift):
    """lanczos.py:5-107: img (b, c, H, W), shift (c, 2) = (dy, dx) per channel; reflect pad 3, vertical then horizontal 7-tap
    correlation with the un-windowed, renormalised Lanczos-3 taps (pi x == 0 -> 1e-6)."""
    import math
    c = img.shape[1]

    def taps(d):
        x = torch.linspace(-3, 3, 7, dtype=d.dtype).view(1, -1) - d.view(-1, 1)
        t = math.pi * x
        t = torch.where(t == 0, torch.tensor(1e-6, dtype=d.dtype), t)
        k = torch.sin(t) / t * torch.sin(t / 3) / (t / 3)
        return k / k.sum(1, keepdim=True)

    ky, kx = taps(shift[:, 0]), taps(shift[:, 1])
    pad = F.pad(img, (3, 3, 3, 3), mode="reflect")
    out = F.conv2d(pad, ky.view(c, 1, 7, 1), groups=c)
    return F.conv2d(out, kx.view(c, 1, 1, 7), groups=c)


def registered_loss_cpsnr(srs, hrs, hr_maps):
    """get_loss(..., metric='cPSNR') of train.py:66-87: brightness bias detached."""
    nclear = torch.sum(hr_maps, dim=(1, 2))
    bright = torch.sum(hr_maps * (hrs - srs), dim=(1, 2)).clone().detach() / nclear
    loss = torch.sum(hr_maps * (srs + bright.view(-1, 1, 1) - hrs) ** 2, dim=(1, 2)) / nclear
    return -10 * torch.log10(loss)


def train_step(lrs, alphas, hrs, hr_maps, keep_mask, hst, sst, lam=1e-6, crop=3):
    """One optimisation step's loss of train.py:172-187 (n_views of the SR = 1) from state dicts that may require grad:
    -> (loss scalar, shifts (B,1,2), srs (B,1,3S,3S), srs_shifted (B,3S,3S))."""
    s3 = 3 * lrs.shape[-1]
    off = (s3 - 128) // 2
    srs = hrnet_forward.__wrapped__(lrs, alphas, hst)
    ref = hrs[:, off:off + 128, off:off + 128].reshape(-1, 1, 128, 128)
    shifts = torch.stack([shiftnet_forward_train(torch.cat([ref, srs[:, :, off:off + 128, off:off + 128]], 1), sst, keep_mask)], 1)
    shifted = lanczos_shift(srs.reshape(-1, 1, s3, s3).transpose(0, 1), shifts.reshape(-1, 2).flip(-1))[:, None].reshape(-1, 1, s3, s3)[:, 0]
    cm = torch.ones((s3, s3), dtype=hrs.dtype)
    cm[:crop] = 0; cm[-crop:] = 0; cm[:, :crop] = 0; cm[:, -crop:] = 0
    loss = -registered_loss_cpsnr(shifted, hrs, cm * hr_maps)
    loss = loss.mean() + lam * shifts.mean() ** 2
    return loss, shifts, srs, shifted
