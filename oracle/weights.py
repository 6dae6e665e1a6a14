"""Portable seeded weights for HRNet / ShiftNet (oracle + tests + goldens).

The generator is numpy PCG64 only (no torch RNG), so the very same tensors can be
re-created on the GPU box without shipping a 137 MB ShiftNet checkpoint or any
reference code.  Key names / shapes are the reference's state-dict contract
(probed from /root/reference/src/DeepNetworks/HRNet.py:36-184 and
ShiftNet.py:9-47).  Values are *not* torch's default init: PReLU slopes and BN
statistics are randomised so that a mis-wired slope / statistic fails parity, and
conv weights are variance preserving so activations stay O(1) through the stack.
"""
import numpy as np

HRNET_SHAPES = [
    ("encode.init_layer.0.weight", (64, 2, 3, 3)),
    ("encode.init_layer.0.bias", (64,)),
    ("encode.init_layer.1.weight", (1,)),
    ("encode.res_layers.0.block.0.weight", (64, 64, 3, 3)),
    ("encode.res_layers.0.block.0.bias", (64,)),
    ("encode.res_layers.0.block.1.weight", (1,)),
    ("encode.res_layers.0.block.2.weight", (64, 64, 3, 3)),
    ("encode.res_layers.0.block.2.bias", (64,)),
    ("encode.res_layers.0.block.3.weight", (1,)),
    ("encode.res_layers.1.block.0.weight", (64, 64, 3, 3)),
    ("encode.res_layers.1.block.0.bias", (64,)),
    ("encode.res_layers.1.block.1.weight", (1,)),
    ("encode.res_layers.1.block.2.weight", (64, 64, 3, 3)),
    ("encode.res_layers.1.block.2.bias", (64,)),
    ("encode.res_layers.1.block.3.weight", (1,)),
    ("encode.final.0.weight", (64, 64, 3, 3)),
    ("encode.final.0.bias", (64,)),
    ("fuse.fuse.0.block.0.weight", (128, 128, 3, 3)),
    ("fuse.fuse.0.block.0.bias", (128,)),
    ("fuse.fuse.0.block.1.weight", (1,)),
    ("fuse.fuse.0.block.2.weight", (128, 128, 3, 3)),
    ("fuse.fuse.0.block.2.bias", (128,)),
    ("fuse.fuse.0.block.3.weight", (1,)),
    ("fuse.fuse.1.weight", (64, 128, 3, 3)),
    ("fuse.fuse.1.bias", (64,)),
    ("fuse.fuse.2.weight", (1,)),
    ("decode.deconv.0.weight", (64, 64, 3, 3)),   # ConvTranspose2d: (Cin, Cout, kH, kW)
    ("decode.deconv.0.bias", (64,)),
    ("decode.deconv.1.weight", (1,)),
    ("decode.final.weight", (1, 64, 1, 1)),
    ("decode.final.bias", (1,)),
]

_SN_CH = [(2, 64), (64, 64), (64, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128)]


def shiftnet_shapes():
    out = []
    for i, (ci, co) in enumerate(_SN_CH, start=1):
        out += [
            (f"layer{i}.0.weight", (co, ci, 3, 3)),
            (f"layer{i}.0.bias", (co,)),
            (f"layer{i}.1.weight", (co,)),
            (f"layer{i}.1.bias", (co,)),
            (f"layer{i}.1.running_mean", (co,)),
            (f"layer{i}.1.running_var", (co,)),
            (f"layer{i}.1.num_batches_tracked", ()),
        ]
    out += [("fc1.weight", (1024, 32768)), ("fc1.bias", (1024,)), ("fc2.weight", (2, 1024))]
    return out


def _fan_in(name, shape):
    if name.startswith("decode.deconv.0"):
        # ConvTranspose2d k3 s3: each output pixel sees Cin inputs through exactly one tap
        return shape[0]
    if len(shape) == 4:
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    return None


def _gen(rng, name, shape, fan_in_of_layer):
    f32 = np.float32
    if name.endswith("num_batches_tracked"):
        return np.array(0, dtype=np.int64)
    if name.endswith("running_mean"):
        return (0.1 * rng.standard_normal(shape)).astype(f32)
    if name.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(f32)
    if len(shape) == 1 and shape[0] == 1 and not name.endswith("bias"):
        # PReLU slope: distinct per module so a swapped slope is caught
        return rng.uniform(0.1, 0.4, shape).astype(f32)
    if len(shape) == 1 and ".1.weight" in name and name.startswith("layer"):
        return rng.uniform(0.5, 1.5, shape).astype(f32)      # BN gamma
    if len(shape) == 1 and ".1.bias" in name and name.startswith("layer"):
        return rng.uniform(-0.2, 0.2, shape).astype(f32)     # BN beta
    if name.endswith("bias"):
        b = 1.0 / np.sqrt(fan_in_of_layer)
        return rng.uniform(-b, b, shape).astype(f32)
    fan_in = _fan_in(name, shape)
    bound = np.sqrt(3.0) * np.sqrt(2.0 / (1.0625 * fan_in))  # variance preserving under PReLU(0.25)
    if name == "fc2.weight":
        bound = 1.0 / 32.0    # reference zero-inits fc2 (ShiftNet.py:47): non-zero here so theta != 0
    if name == "fc1.weight":
        # 33.5 M values: generate in float32 directly to keep the generator cheap
        return ((rng.random(shape, dtype=f32) * 2.0 - 1.0) * f32(bound)).astype(f32)
    return rng.uniform(-bound, bound, shape).astype(f32)


def _state(shapes, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    last_fan = 1
    for name, shape in shapes:
        fi = _fan_in(name, shape)
        if fi is not None and name.endswith("weight"):
            last_fan = fi
        out[name] = _gen(rng, name, shape, last_fan)
    return out


def hrnet_state(seed=1234):
    """name -> float32 ndarray with the reference HRNet state-dict keys/shapes."""
    return _state(HRNET_SHAPES, seed)


def shiftnet_state(seed=4321):
    """name -> ndarray with the reference ShiftNet state-dict keys/shapes (fc2 non-zero)."""
    return _state(shiftnet_shapes(), seed)


def to_torch_state(state):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}


HRNET_CONFIG = {   # == /root/reference/config/config.json:8-34 ("network")
    "encoder": {"in_channels": 2, "num_layers": 2, "kernel_size": 3, "channel_size": 64},
    "recursive": {"alpha_residual": True, "in_channels": 64, "num_layers": 2, "kernel_size": 3},
    "decoder": {
        "deconv": {"in_channels": 64, "kernel_size": 3, "stride": 3, "out_channels": 64},
        "final": {"in_channels": 64, "kernel_size": 1, "out_channels": 1},
    },
}
