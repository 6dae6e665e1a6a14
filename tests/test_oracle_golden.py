"""CPU: the numpy oracle (oracle/hrnet_np.py) against the fixtures produced by the reference itself
(oracle/make_goldens.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path
with the oracle and with the same fixtures."""
import os

import numpy as np
import pytest

from oracle import hrnet_np as O
from oracle import synth, weights

HST = weights.hrnet_state(1234)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


HR_CASES = ["hrnet_b2_v5_s16", "hrnet_b1_v1_s16", "hrnet_b1_v12_s24", "hrnet_b2_v6_s16_pad",
            "hrnet_b2_v4_s16_noalpha", "hrnet_b1_v32_s32"]


@pytest.mark.parametrize("name", HR_CASES)
def test_hrnet_forward_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    stages = {}
    sr = O.hrnet_forward(g["lrs"], g["alphas"], HST, alpha_residual=bool(g["alpha_residual"]), stages=stages)
    assert sr.shape == g["sr"].shape
    assert rel_err(sr, g["sr"]) < 2e-5            # fp32 reference vs fp64 oracle
    if "emb" in g.files:
        assert np.array_equal(stages["ref"].astype(np.float32), g["ref"])       # median is exact
        assert rel_err(stages["emb"], g["emb"]) < 2e-5
        assert rel_err(stages["fused"], g["fused"]) < 2e-5


def test_hrnet_c1_shape_matches_reference(golden_dir):
    """BASELINE config 1 shape (B=4, V=4, 128->384); inputs are re-synthesised from the seed."""
    g = load(golden_dir, "hrnet_c1_b4_v4_s128")
    b, v, s = (int(x) for x in g["shape"])
    lrs, alphas, _ = synth.make_batch(int(g["seed"]), b, v, s, [int(x) for x in g["n_real"]])
    assert abs(lrs.astype(np.float64).sum() - float(g["lrs_sum"])) < 1e-6      # synth is portable
    sr = O.hrnet_forward(lrs, alphas, HST, dtype=np.float32)
    assert rel_err(sr, g["sr"]) < 1e-4


def test_lower_median_and_pairing():
    x = np.array([4.0, 1.0, 3.0, 2.0]).reshape(1, 4, 1, 1)
    assert O.reference_frame(x)[0, 0, 0] == 2.0      # lower middle of an even count
    x = np.arange(12, dtype=np.float64).reshape(1, 12, 1, 1)
    assert O.reference_frame(x)[0, 0, 0] == 4.0      # only the first 9 views enter
    # pairing i <-> n-parity-1-i with the odd leftover dropped: identity-ish fuse by zero weights
    st = {k: np.zeros_like(v) for k, v in HST.items()}
    emb = np.arange(5, dtype=np.float64).reshape(1, 5, 1, 1, 1) * np.ones((1, 5, 64, 2, 2))
    out, al = O.fuse_level(emb, np.ones((1, 5)), st)
    assert out.shape[1] == 2 and np.allclose(out[0, :, 0, 0, 0], [0, 1])    # alice kept, f == 0


def test_shiftnet_eval_matches_reference(golden_dir):
    g = load(golden_dir, "shiftnet_eval_b3")
    sst = weights.shiftnet_state(4321)
    layers = []
    theta = O.shiftnet_forward(g["x"], sst, layers_out=layers)
    assert rel_err(theta, g["theta"]) < 1e-4
    for i, l in enumerate(layers, start=1):
        assert rel_err(l[:, ::8, ::4, ::4], g[f"layer{i}_sample"]) < 1e-4


def test_shiftnet_train_bn_matches_reference(golden_dir):
    g = load(golden_dir, "shiftnet_train_b4")
    sst = weights.shiftnet_state(4321)
    mask = np.unpackbits(g["dropout_mask"], axis=1)[:, :32768]
    layers = []
    theta = O.shiftnet_forward(g["x"], sst, train_bn=True, dropout_mask=mask, layers_out=layers)
    for i, l in enumerate(layers, start=1):
        assert rel_err(l[:, ::8, ::4, ::4], g[f"layer{i}_sample"]) < 1e-4
    assert rel_err(theta, g["theta"]) < 1e-4


def test_lanczos_matches_reference(golden_dir):
    g = load(golden_dir, "lanczos")
    taps = O.lanczos_kernel(g["d"])
    assert np.abs(taps - g["taps"]).max() < 2e-6
    assert np.abs(taps.sum(axis=1) - 1).max() < 1e-6
    out = O.lanczos_shift(g["img"], g["shift"], p=3)
    assert np.abs(out - g["shifted"]).max() < 5e-6
    assert np.abs(O.lanczos_shift(g["img"], g["shift"], p=5) - out).max() == 0     # independent of p >= 3
    tr = O.shiftnet_transform(g["theta"], g["imgs"])
    assert tr.shape == g["transformed"].shape == (1, 1, 5, 48, 48)
    assert np.abs(tr - g["transformed"]).max() < 5e-6
    # zero shift is the identity up to tap rounding
    assert np.abs(out[:, 0] - g["img"][:, 0]).max() < 1e-5


def test_callers_match_reference(golden_dir):
    g = load(golden_dir, "callers")
    mask = g["maps"] * g["crop"][0]
    assert rel_err(O.get_loss(g["srs"], g["hrs"], mask, "cPSNR"), g["loss_cpsnr"]) < 1e-5
    assert rel_err(O.get_loss(g["srs"], g["hrs"], mask, "cMSE"), g["loss_cmse"]) < 1e-5
    srn = np.clip(g["srs"], 0, 1)
    assert rel_err(O.cpsnr(srn, g["hrs"], g["maps"]), g["cpsnr"]) < 1e-9
    assert rel_err(O.shift_cpsnr(srn, g["hrs"], g["maps"]), g["shift_cpsnr"]) < 1e-9
    # registration glue: ShiftNet (eval restated) + Lanczos on the reference's SR output
    sst = weights.shiftnet_state(4321)
    off = (144 - 128) // 2
    pairs = np.stack([g["hrs2"][:, off:off + 128, off:off + 128], g["srs2"][:, 0, off:off + 128, off:off + 128]], 1)
    theta = O.shiftnet_forward(pairs, sst)
    assert rel_err(theta, g["shifts"][:, 0]) < 1e-4
    shifted = O.shiftnet_transform(g["shifts"][:, 0], g["srs2"])[0, 0]
    assert np.abs(shifted - g["shifted2"]).max() < 2e-5 * max(1.0, np.abs(g["shifted2"]).max())


@pytest.mark.parametrize("name", ["hrnet_b2_v5_s16", "hrnet_b2_v4_s16_noalpha", "hrnet_b1_v32_s32"])
def test_torch_cpu_port_matches_reference(golden_dir, name):
    """oracle/torch_port.py (the cpu_baseline of bench.py) computes the same forward as the reference."""
    import torch
    from oracle import torch_port
    g = load(golden_dir, name)
    st = weights.to_torch_state(HST)
    sr = torch_port.hrnet_forward(torch.from_numpy(g["lrs"]), torch.from_numpy(g["alphas"]), st,
                                  alpha_residual=bool(g["alpha_residual"])).numpy()
    assert rel_err(sr, g["sr"]) < 1e-5


def test_train_step_port_matches_reference(golden_dir):
    """The torch port the gradient tests use as their autograd oracle (oracle/torch_port.train_step, fp64) against ONE FULL TRAIN
    STEP OF THE REFERENCE ITSELF (oracle/make_goldens.py::train_step_golden: train.py:164-190 on the reference's modules, fp64,
    hooked dropout mask): loss, shifts, SR crops and every parameter gradient (L2 norm, sum, strided sample)."""
    import torch
    from oracle import torch_port
    g = load(golden_dir, "train_step")
    B, V, S = (int(v) for v in g["shape"])
    lrs, alphas, hrs = synth.make_batch(31, B, V, S, V)
    rng = np.random.Generator(np.random.PCG64(5))
    maps = (rng.random((B, 3 * S, 3 * S)) > 0.1).astype(np.float32)
    keep = (rng.random((B, 32768)) >= 0.5)
    hst = {k: v.double().requires_grad_(True) for k, v in weights.to_torch_state(weights.hrnet_state(1234)).items()}
    sst = {k: v.double().requires_grad_("running" not in k and "num_batches" not in k) for k, v in weights.to_torch_state(weights.shiftnet_state(4321)).items()}
    torch.set_num_threads(8)
    with torch.enable_grad():
        loss, shifts, srs, shifted = torch_port.train_step(torch.from_numpy(lrs).double(), torch.from_numpy(alphas).double(),
                                                          torch.from_numpy(hrs).double(), torch.from_numpy(maps).double(),
                                                          torch.from_numpy(keep).double(), hst, sst, lam=float(g["lam"]), crop=int(g["crop"]))
        loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-8 * abs(float(g["loss"]))      # fp64 conv blocking differs with the thread count
    assert rel_err(shifts.detach().numpy(), g["shifts"]) <= 1e-9
    assert rel_err(srs.detach().numpy()[:, :, 40:72, 40:72], g["srs_crop"]) <= 1e-9
    assert rel_err(shifted.detach().numpy()[:, 40:72, 40:72], g["srs_shifted_crop"]) <= 1e-9
    for prefix, st in (("hrnet", hst), ("shiftnet", sst)):
        for k, v in st.items():
            if not v.requires_grad:
                continue
            got = v.grad.numpy().ravel()
            stride = int(g[f"{prefix}/{k}/stride"])
            scale = max(float(g[f"{prefix}/{k}/absmax"]), 1e-300)
            if prefix == "shiftnet" and k.endswith(".0.bias"):
                continue                # conv bias in front of a train-mode BatchNorm: a mathematically zero gradient (rounding noise)
            tol = 1e-7 * scale
            if f"{prefix}/{k}/abs_terms" in g.files:       # a cancelling sum (decode.final.bias: exactly zero by the brightness correction)
                tol = 1e-9 * float(g[f"{prefix}/{k}/abs_terms"])
            assert np.abs(got[::stride] - g[f"{prefix}/{k}/sample"]).max() <= tol, (prefix, k)
            if f"{prefix}/{k}/abs_terms" in g.files:
                continue
            assert abs(np.sqrt((got * got).sum()) - float(g[f"{prefix}/{k}/norm"])) <= 1e-7 * max(float(g[f"{prefix}/{k}/norm"]), 1e-300), (prefix, k)
