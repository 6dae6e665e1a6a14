"""Helpers shared by the GPU parity tests."""
import os

import numpy as np
import torch

from oracle import weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def psnr_db(a, b):
    """10 log10(peak^2 / mse) with peak = max|b|: a scale-free agreement measure for the bf16 path."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = ((a - b) ** 2).mean()
    return float(10 * np.log10(max(np.abs(b).max(), 1e-30) ** 2 / max(mse, 1e-300)))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


_models = {}


def hip_hrnet(precision="fp32", alpha_residual=True, seed=1234):
    from DeepNetworks.HRNet import HRNet
    key = (precision, alpha_residual, seed)
    if key not in _models:
        cfg = {k: dict(v) for k, v in weights.HRNET_CONFIG.items()}
        cfg["recursive"]["alpha_residual"] = alpha_residual
        m = HRNet(cfg)
        m.load_state_dict(weights.to_torch_state(weights.hrnet_state(seed)))
        m.precision = precision
        _models[key] = m.cuda().eval()
    return _models[key]


def hip_shiftnet(seed=4321):
    from DeepNetworks.ShiftNet import ShiftNet
    m = ShiftNet()
    m.load_state_dict(weights.to_torch_state(weights.shiftnet_state(seed)))
    return m.cuda().eval()


def nhwc_to_nchw(t, prec=None):
    """(..., H, W, C) storage tensor -> float32 numpy (..., C, H, W); bf16x3 stage tensors are (2, ...) planes: hi + lo."""
    if prec == "bf16x3":
        t = t[0].float() + t[1].float()
    nd = t.dim()
    perm = list(range(nd - 3)) + [nd - 1, nd - 3, nd - 2]
    return t.float().permute(*perm).contiguous().cpu().numpy()
