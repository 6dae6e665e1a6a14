"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes -> libhrnet_hip.so), against
  (1) the fixtures produced by the reference itself (tests/golden, oracle/make_goldens.py),
  (2) the numpy oracle on fresh seeded inputs (edge shapes), and
  (3) size-independent properties at BASELINE.json's full sizes.

Tolerances (BASELINE.json north_star: "within 1e-3 rel fp32"):
  fp32 path : max|hip - ref| / max|ref| <= 1e-3 is the contract; the exact-fp32 MFMA path is additionally held to
              2e-5 as a regression guard (measured ~2e-6).
  bf16 path : bf16 storage cannot meet 1e-3; it is held to max-rel <= 2.5e-2 and PSNR(hip, ref) >= 45 dB (measured over
              the checks of this file: worst 1.35e-2 / 49.5 dB, typically 1e-2 / 52-56 dB; HRN_TEST_RECORD=file lists them)
              and documented as such in DESIGN.md.
"""
import os

import numpy as np
import pytest
import torch

from oracle import hrnet_np as O
from oracle import synth, weights
import util

pytestmark = pytest.mark.gpu

FP32_CONTRACT, FP32_GUARD = 1e-3, 2e-5
BF16_REL, BF16_PSNR = 2.5e-2, 45.0
X3_REL = 1e-4          # bf16x3 (split-bf16, three MFMAs per product): inside the 1e-3 contract by 10x; measured worst ~2e-5

HR_CASES = ["hrnet_b1_v1_s16", "hrnet_b2_v5_s16", "hrnet_b2_v6_s16_pad", "hrnet_b1_v12_s24", "hrnet_b2_v4_s16_noalpha",
            "hrnet_b1_v32_s32"]


def _check(prec, got, want):
    if prec == "fp32":
        e = util.rel_err(got, want)
        assert e <= FP32_CONTRACT and e <= FP32_GUARD, e
    elif prec == "bf16x3":
        e = util.rel_err(got, want)
        if os.environ.get("HRN_TEST_RECORD"):
            with open(os.environ["HRN_TEST_RECORD"], "a") as f:
                f.write(f"bf16x3 {e:.4e}\n")
        assert e <= FP32_CONTRACT and e <= X3_REL, e
    else:
        if os.environ.get("HRN_TEST_RECORD"):      # measured margins of the bf16 bounds: one line per check
            with open(os.environ["HRN_TEST_RECORD"], "a") as f:
                f.write(f"{util.rel_err(got, want):.4e} {util.psnr_db(got, want):.2f}\n")
        assert util.rel_err(got, want) <= BF16_REL and util.psnr_db(got, want) >= BF16_PSNR, (util.rel_err(got, want), util.psnr_db(got, want))


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("name", HR_CASES)
def test_hrnet_forward_vs_reference_golden(name, prec):
    g = util.golden(name)
    m = util.hip_hrnet(prec, bool(g["alpha_residual"]))
    with torch.no_grad():
        sr = m(util.dev(g["lrs"]), util.dev(g["alphas"]))
    assert sr.shape == g["sr"].shape and sr.dtype == torch.float32
    _check(prec, sr.cpu().numpy(), g["sr"])


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("name", ["hrnet_b2_v5_s16", "hrnet_b2_v6_s16_pad", "hrnet_b1_v1_s16"])
def test_hrnet_stages_vs_reference_golden(name, prec):
    g = util.golden(name)
    m = util.hip_hrnet(prec)
    lrs, alphas = util.dev(g["lrs"]), util.dev(g["alphas"])
    with torch.no_grad():
        emb = m.encode_views(lrs)
        assert tuple(emb.shape[-5:]) == g["lrs"].shape + (64,)
        _check(prec, util.nhwc_to_nchw(emb, prec), g["emb"])
        fused = m.fuse_views(emb, alphas)
        _check(prec, util.nhwc_to_nchw(fused, prec), g["fused"])
        from hrnet_hip import binding
        gf = binding.float_to_planes(torch.from_numpy(g["fused"]).cuda().permute(0, 2, 3, 1).contiguous(), m._dtype())
        _check(prec, m.decode_state(gf).cpu().numpy(), g["sr"])


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
def test_hrnet_config1_shape(prec):
    """BASELINE config 1 (B=4, V=4, 128->384) against the reference's own output."""
    g = util.golden("hrnet_c1_b4_v4_s128")
    b, v, s = (int(x) for x in g["shape"])
    lrs, alphas, _ = synth.make_batch(int(g["seed"]), b, v, s, [int(x) for x in g["n_real"]])
    with torch.no_grad():
        sr = util.hip_hrnet(prec)(util.dev(lrs), util.dev(alphas)).cpu().numpy()
    _check(prec, sr, g["sr"])


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("shape", [(1, 2, 20), (3, 3, 8), (1, 7, 40), (2, 9, 33), (1, 16, 12), (1, 4, 72)])
def test_hrnet_edge_shapes_vs_oracle(shape, prec):
    """Sizes that do not fill the 8x32 / 16x32 tiles, odd view counts, tiny images, and one image (72) that spans more than two
    tiles in both directions without being a tile multiple: the fp32 AND the bf16 path vs the fp64 numpy oracle."""
    b, v, s = shape
    lrs, alphas, _ = synth.make_batch(1000 + s, b, v, s, [max(1, v - i) for i in range(b)])
    want = O.hrnet_forward(lrs, alphas, weights.hrnet_state(1234))
    with torch.no_grad():
        sr = util.hip_hrnet(prec)(util.dev(lrs), util.dev(alphas)).cpu().numpy()
    _check(prec, sr, want)


def test_general_conv_kernel_route_vs_reference_golden():
    """The bf16 layers that conv3x3_r64 / conv3x3_v6 decline (images beyond their 32-bit in-image offsets) run on the general kernel
    of conv3x3.hip.  No BASELINE config is that large, so the route is forced (HRN_CONV_R64=0 HRN_CONV_V6=0, read once per process:
    a fresh interpreter started before any GPU call) and checked against the reference's own outputs, ragged sizes included."""
    import subprocess
    import sys
    code = (
        "import sys, os, numpy as np, torch\n"
        f"sys.path[:0] = [{os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}, {os.path.dirname(os.path.abspath(__file__))!r}]\n"
        "import conftest, util\n"
        "from oracle import hrnet_np as O, synth, weights\n"
        "worst = 0.0\n"
        "for name in ('hrnet_b2_v5_s16', 'hrnet_b1_v12_s24', 'hrnet_b1_v32_s32'):\n"
        "    g = util.golden(name)\n"
        "    with torch.no_grad():\n"
        "        sr = util.hip_hrnet('bf16')(util.dev(g['lrs']), util.dev(g['alphas'])).cpu().numpy()\n"
        "    e, ps = util.rel_err(sr, g['sr']), util.psnr_db(sr, g['sr'])\n"
        "    assert e <= 2.5e-2 and ps >= 45.0, (name, e, ps)\n"
        "lrs, alphas, _ = synth.make_batch(1072, 1, 4, 72, [4])\n"
        "want = O.hrnet_forward(lrs, alphas, weights.hrnet_state(1234))\n"
        "with torch.no_grad():\n"
        "    sr = util.hip_hrnet('bf16')(util.dev(lrs), util.dev(alphas)).cpu().numpy()\n"
        "e, ps = util.rel_err(sr, want), util.psnr_db(sr, want)\n"
        "assert e <= 2.5e-2 and ps >= 45.0, ('72', e, ps)\n"
        "from hrnet_hip import binding\n"
        "binding.profile_enable(True)\n"
        "with torch.no_grad():\n"
        "    util.hip_hrnet('bf16')(util.dev(lrs), util.dev(alphas))\n"
        "torch.cuda.synchronize(); binding.profile_enable(False)\n"
        "print('FAMILIES', sorted(binding.profile_read()))\n"
    )
    env = dict(os.environ, HRN_CONV_R64="0", HRN_CONV_V6="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    # the general kernel files every layer under the plain family names; r64 / v6 would have filed their residual layers under "...+res"
    assert "conv3x3_bf16_128x128" in r.stdout and "conv3x3_bf16_64x64" in r.stdout and "+res" not in r.stdout


def test_hrnet_other_weights_and_layers():
    """A second weight seed and a non-default encoder depth (num_layers=3 and 0) vs the oracle."""
    from DeepNetworks.HRNet import HRNet
    for nl in (0, 3):
        cfg = {k: dict(v) for k, v in weights.HRNET_CONFIG.items()}
        cfg["encoder"]["num_layers"] = nl
        m = HRNet(cfg).cuda().eval()
        st = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        lrs, alphas, _ = synth.make_batch(77, 2, 4, 16)
        want = O.hrnet_forward(lrs, alphas, st, num_layers=nl)
        with torch.no_grad():
            sr = m(util.dev(lrs), util.dev(alphas)).cpu().numpy()
        assert util.rel_err(sr, want) <= FP32_GUARD


def test_packed_parameters_follow_updates():
    m = util.hip_hrnet("fp32", seed=99)
    lrs, alphas, _ = synth.make_batch(5, 1, 2, 16)
    with torch.no_grad():
        a = m(util.dev(lrs), util.dev(alphas)).clone()
        m.decode.final.bias.add_(1.0)                    # in-place update bumps the version -> re-pack
        b = m(util.dev(lrs), util.dev(alphas))
        m.decode.final.bias.sub_(1.0)
    assert torch.allclose(b, a + 1.0, atol=1e-5)


# ----------------------------------------------------------------------------- full-size properties (BASELINE configs 2/3)
@pytest.mark.parametrize("prec,b,v", [("bf16", 32, 32), ("fp32", 16, 16), ("bf16x3", 32, 32)])
def test_full_size_properties(prec, b, v):
    """At the metric's size the oracle is too slow; check properties that must hold bit-exactly:
    determinism, batch independence (a sample alone == the sample inside the batch) and invariance to the content
    of padded views (alpha = 0, index >= 9 so they are outside the median window and never an 'alice')."""
    lrs, alphas = synth.fast_batch(3, b, v, 128)
    n_real = v - 5
    alphas[:, n_real:] = 0.0
    lrs[:, n_real:] = 0.0
    m = util.hip_hrnet(prec)
    x, a = util.dev(lrs), util.dev(alphas)
    with torch.no_grad():
        y0 = m(x, a).clone()
        y1 = m(x, a).clone()
        assert torch.equal(y0, y1)
        assert torch.isfinite(y0).all()
        one = m(x[5:6].contiguous(), a[5:6].contiguous())
        assert torch.equal(one[0], y0[5])
        x2 = x.clone()
        x2[:, n_real:] = torch.rand_like(x2[:, n_real:])
        y2 = m(x2, a)
        assert torch.equal(y2, y0)
        # alpha really gates: switching the padded views on changes the output
        a2 = a.clone()
        a2[:, n_real:] = 1.0
        assert not torch.equal(m(x2, a2), y0)
    # a sample of the full-size batch against the oracle would take minutes; a 1-sample fp32 check at V=16 is affordable
    if prec == "fp32":
        want = O.hrnet_forward(lrs[:1], alphas[:1], weights.hrnet_state(1234), dtype=np.float32)
        assert util.rel_err(y0[:1].cpu().numpy(), want) <= 1e-4


def test_config5_full_size_properties():
    """BASELINE configs[4] at its full size: B=32, n_views=32, 512x512 -> 1536x1536, bf16 (1 GiB of input, 96 GiB of workspace on
    the 288 GB part).  The oracle cannot run this; the bit-exact properties of `test_full_size_properties` can: determinism,
    batch independence, invariance to the content of padded views, and alpha really gating."""
    b, v, s = 32, 32, 512
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip(f"needs ~100 GiB of device memory, {free / 2 ** 30:.0f} GiB free")
    lrs, alphas = synth.fast_batch(55, b, v, s)
    n_real = v - 5
    alphas[:, n_real:] = 0.0
    lrs[:, n_real:] = 0.0
    m = util.hip_hrnet("bf16")
    x, a = util.dev(lrs), util.dev(alphas)
    del lrs
    with torch.no_grad():
        y0 = m(x, a).clone()
        assert y0.shape == (b, 1, 3 * s, 3 * s) and bool(torch.isfinite(y0).all())
        assert torch.equal(m(x, a), y0)                                         # deterministic
        one = m(x[7:8].contiguous(), a[7:8].contiguous())
        assert torch.equal(one[0], y0[7])                                       # a sample alone == the sample in the batch
        x[:, n_real:] = torch.rand_like(x[:, n_real:])
        assert torch.equal(m(x, a), y0)                                         # padded views (alpha 0) do not matter
        a[:, n_real:] = 1.0
        assert not torch.equal(m(x, a), y0)                                     # ... because alpha gates them
    from hrnet_hip import binding
    binding._ws_cache.clear()                                                   # hand the 96 GiB back before the next test
    torch.cuda.empty_cache()


def test_config5_image_size_fp32_vs_oracle():
    """The fp32 path at configs[4]'s image size against the oracle itself (one sample, two views: what the oracle does in
    ~20 s), and the bf16 path against the same oracle output: the 512 x 512 tiling (32 x 16 tiles per image, in-image byte
    offsets up to 64 MB) is checked against the reference's arithmetic, not only against our own other path."""
    lrs, alphas = synth.fast_batch(77, 1, 2, 512)
    want = O.hrnet_forward(lrs, alphas, weights.hrnet_state(1234), dtype=np.float32)
    x, a = util.dev(lrs), util.dev(alphas)
    with torch.no_grad():
        got32 = util.hip_hrnet("fp32")(x, a).cpu().numpy()
        got16 = util.hip_hrnet("bf16")(x, a).cpu().numpy()
        gotx3 = util.hip_hrnet("bf16x3")(x, a).cpu().numpy()
    assert got32.shape == want.shape == (1, 1, 1536, 1536)
    assert util.rel_err(got32, want) <= 1e-4
    _check("bf16", got16, want)
    _check("bf16x3", gotx3, want)


# ----------------------------------------------------------------------------- ShiftNet
def test_shiftnet_eval_vs_reference_golden():
    g = util.golden("shiftnet_eval_b3")
    m = util.hip_shiftnet()
    with torch.no_grad():
        th = m(util.dev(g["x"]))
    assert th.shape == (3, 2)
    assert util.rel_err(th.cpu().numpy(), g["theta"]) <= 1e-4


def test_shiftnet_train_mode_vs_reference_golden():
    """Train mode: batch-statistics BatchNorm + the reference run's dropout mask (recovered by make_goldens)."""
    from hrnet_hip import binding
    g = util.golden("shiftnet_train_b4")
    m = util.hip_shiftnet().train()
    mask = np.unpackbits(g["dropout_mask"], axis=1)[:, :32768].astype(np.uint8)
    with torch.no_grad():
        th = binding.shiftnet_forward(m.packed_parameters(), m._named(), util.dev(g["x"]), train_bn=True, momentum=0.1,
                                      dropout_mask=util.dev(mask))
    assert util.rel_err(th.cpu().numpy(), g["theta"]) <= 1e-3       # measured 7e-6
    for i in range(1, 9):
        bn = getattr(m, f"layer{i}")[1]
        assert np.abs(bn.running_mean.cpu().numpy() - g[f"layer{i}_running_mean"]).max() < 1e-5
        assert np.abs(bn.running_var.cpu().numpy() - g[f"layer{i}_running_var"]).max() < 1e-5


def test_shiftnet_module_train_mode_runs_and_zero_init():
    from DeepNetworks.ShiftNet import ShiftNet
    m = ShiftNet().cuda().train()
    x = torch.rand(4, 2, 128, 128, device="cuda")
    with torch.no_grad():
        th = m(x)
    assert torch.equal(th, torch.zeros_like(th))          # fc2 is zero-initialised (ShiftNet.py:47)
    assert int(m.layer3[1].num_batches_tracked) == 1
    with pytest.raises(ValueError):
        with torch.no_grad():
            m(torch.rand(1, 2, 64, 64, device="cuda"))


# ----------------------------------------------------------------------------- Lanczos
def test_lanczos_vs_reference_golden():
    import lanczos
    g = util.golden("lanczos")
    taps = lanczos.lanczos_kernel(util.dev(g["d"]))
    assert taps.shape == (9, 7)
    assert np.abs(taps.cpu().numpy() - g["taps"]).max() <= 2e-6
    out = lanczos.lanczos_shift(util.dev(g["img"]), util.dev(g["shift"]), p=3)
    assert np.abs(out.cpu().numpy() - g["shifted"]).max() <= 5e-6
    tr = util.hip_shiftnet().transform(util.dev(g["theta"]), util.dev(g["imgs"]))
    assert tuple(tr.shape) == (1, 1, 5, 48, 48)
    assert np.abs(tr.cpu().numpy() - g["transformed"]).max() <= 5e-6
    with pytest.raises(NotImplementedError):
        lanczos.lanczos_shift(util.dev(g["img"]), util.dev(g["shift"]), a=2)


def test_lanczos_full_size_properties():
    """B=32 SR frames of 384x384 (the training-step shape): zero shift is the identity (to tap rounding), an integer
    shift is a pure translation away from the border, and the result matches the oracle on one frame."""
    import lanczos
    rng = np.random.Generator(np.random.PCG64(8))
    img = rng.random((1, 32, 384, 384), dtype=np.float32)
    shift = np.zeros((32, 2), np.float32)
    shift[1] = (2.0, -1.0)
    shift[2] = (0.37, -1.6)
    out = lanczos.lanczos_shift(util.dev(img), util.dev(shift)).cpu().numpy()
    assert np.abs(out[0, 0] - img[0, 0]).max() < 1e-5
    assert np.abs(out[0, 1, 8:-8, 8:-8] - img[0, 1, 10:-6, 7:-9]).max() < 1e-5       # out(y,x) = in(y+dy, x+dx)
    want = O.lanczos_shift(img[:, 2:3], shift[2:3])
    assert np.abs(out[:, 2:3] - want).max() < 5e-6


def test_registration_glue_vs_reference_golden():
    """register_batch + apply_shifts (train.py:26-63) re-enacted on the new modules, on the reference's own SR output."""
    g = util.golden("callers")
    sn = util.hip_shiftnet()
    srs = util.dev(g["srs2"])
    hrs = util.dev(g["hrs2"])
    off = (144 - 128) // 2
    with torch.no_grad():
        pairs = torch.cat([hrs[:, off:off + 128, off:off + 128].reshape(-1, 1, 128, 128), srs[:, :, off:off + 128, off:off + 128]], 1)
        thetas = torch.stack([sn(pairs)], 1)                                   # (B, n_views=1, 2)
        images = srs.view(-1, 1, 144, 144)
        new = sn.transform(thetas.view(-1, 2), images, device="cuda").view(-1, 1, 144, 144)[:, 0]
    assert util.rel_err(thetas.cpu().numpy(), g["shifts"]) <= 1e-4
    assert np.abs(new.cpu().numpy() - g["shifted2"]).max() <= 2e-5 * max(1.0, np.abs(g["shifted2"]).max())


# ----------------------------------------------------------------------------- error behaviour
def test_errors_are_loud():
    m = util.hip_hrnet("fp32")
    with pytest.raises(ValueError):
        m(torch.zeros(1, 2, 8, 16, device="cuda"), torch.ones(1, 2, device="cuda"))      # non-square
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 8, 8), torch.ones(1, 2))                                      # CPU tensors: no fallback
    m2 = util.hip_hrnet("fp32", seed=5)
    m2.precision = "fp8"
    with pytest.raises(ValueError):
        m2(torch.zeros(1, 2, 8, 8, device="cuda"), torch.ones(1, 2, device="cuda"))
    m2.precision = "fp32"


@pytest.mark.parametrize("S,V", [(8, 2), (18, 2), (20, 3), (27, 3), (40, 5), (50, 2), (72, 4), (100, 2)])
def test_bf16_kernels_on_awkward_sizes_vs_fp32_path(S, V):
    """Image sides that are no multiple of any tile (conv3x3_v6: 16 x 32, conv3x3_r64 / v3: 8 x 32; LDS-DMA halo pieces that end
    mid-row, EXEC-masked last pieces, partial store rows; widths that split a lane pair (27) or a lane quad (18, 50) of the
    coalescing lane exchanges in the epilogues): the bf16 kernels against the exact-fp32 path on the same inputs."""
    lrs, alphas = synth.fast_batch(40 + S, 2, V, S)
    x, a = util.dev(lrs), util.dev(alphas)
    with torch.no_grad():
        ref = util.hip_hrnet("fp32")(x, a).cpu().numpy()
        got = util.hip_hrnet("bf16")(x, a).cpu().numpy()
        again = util.hip_hrnet("bf16")(x, a).cpu().numpy()
    assert np.isfinite(got).all() and np.array_equal(got, again)
    assert util.rel_err(got, ref) <= 4e-2 and util.psnr_db(got, ref) >= 42.0


def test_entry_points_are_registered_torch_ops():
    """north_star: "exposed to Python as PyTorch-ROCm custom ops".  Inference AND training entry points are dispatcher-registered
    (torch.ops.hrnet_hip.*) with fake implementations and - where train.py differentiates through them - autograd formulas that are
    registered ops themselves: callable through torch.ops, shape-inferable on the meta device, `torch.library.opcheck` clean, and the
    modules go through them in both modes."""
    from hrnet_hip import binding
    lrs, alphas = synth.fast_batch(5, 2, 4, 32)
    m = util.hip_hrnet("fp32")
    packed, dt = m.packed_parameters()
    x, a = util.dev(lrs), util.dev(alphas)
    with torch.no_grad():
        via_ops = torch.ops.hrnet_hip.hrnet_forward(packed, dt, 2, True, x, a)
        assert torch.equal(via_ops, m(x, a))
        img = torch.rand(1, 3, 40, 40, device="cuda")
        sh = torch.tensor([[0.3, -0.2], [0.0, 0.0], [1.5, 0.25]], device="cuda")
        assert torch.equal(torch.ops.hrnet_hip.lanczos_shift(img, sh), binding.lanczos_shift(img, sh))
    basic = ("test_schema", "test_faketensor")
    full = ("test_schema", "test_faketensor", "test_autograd_registration")
    ops = torch.ops.hrnet_hip
    for name in ("hrnet_forward", "lanczos_shift", "lanczos_kernel", "shift_cpsnr", "hrnet_forward_train", "hrnet_backward",
                 "shiftnet_forward_train", "shiftnet_backward", "lanczos_shift_backward", "get_loss_train", "get_loss_backward", "adam_step"):
        assert hasattr(ops, name), name
    torch.library.opcheck(ops.lanczos_shift.default, (img.clone().requires_grad_(True), sh.clone().requires_grad_(True)), test_utils=full)
    torch.library.opcheck(ops.lanczos_shift_backward.default, (img, sh, torch.rand_like(img)), test_utils=basic)
    # HRNet training pair
    mt = util.hip_hrnet("fp32", seed=77).train()
    params = [p for _, p in mt.named_parameters()]
    p32 = mt._packed_f32()
    torch.library.opcheck(ops.hrnet_forward_train.default, (p32, x, a, params, 2, True, binding.F32), test_utils=full)
    sr, tws = ops.hrnet_forward_train(p32, x, a, params, 2, True, binding.F32)
    torch.library.opcheck(ops.hrnet_backward.default, (p32, [p.detach() for p in params], x, a, torch.rand_like(sr), tws, 2, True, binding.F32), test_utils=basic)
    mt.eval()
    # ShiftNet training pair
    from DeepNetworks.ShiftNet import ShiftNet
    sn = ShiftNet().cuda().train()
    with torch.no_grad():
        sn.fc2.weight.normal_(0.0, 1e-3)
    named = sn._named()
    sp = [named[k] for k in binding.SHIFTNET_PARAM_NAMES]
    sb = [named[k] for k in binding.SHIFTNET_BUFFER_NAMES]
    pairs = torch.rand(2, 2, 128, 128, device="cuda")
    mask = (torch.rand(2, 32768, device="cuda") >= 0.5).to(torch.uint8)
    before = [b.clone() for b in sb]
    torch.library.opcheck(ops.shiftnet_forward_train.default, (sn.packed_parameters(), pairs.clone().requires_grad_(True), sp, sb, 0.1, mask), test_utils=full)
    assert all(torch.equal(b0, b1) for b0, b1 in zip(before, sb))                    # functional: the op itself leaves the buffers alone
    theta, stws, new_running = ops.shiftnet_forward_train(sn.packed_parameters(), pairs, sp, sb, 0.1, mask)
    assert not torch.equal(new_running[0], sb[0])
    torch.library.opcheck(ops.shiftnet_backward.default, ([p.detach() for p in sp], pairs, mask, torch.rand_like(theta), stws, True), test_utils=basic)
    theta_mod = sn(pairs)                                                             # the module copies the new statistics into its buffers
    assert theta_mod.requires_grad and int(sn.layer1[1].num_batches_tracked) == 1 and not torch.equal(before[0], sn.layer1[1].running_mean)
    # registered loss
    srs, hrs = torch.rand(2, 48, 48, device="cuda"), torch.rand(2, 48, 48, device="cuda")
    maps = (torch.rand(2, 48, 48, device="cuda") > 0.2).float()
    torch.library.opcheck(ops.get_loss_train.default, (srs.clone().requires_grad_(True), hrs, maps, "cPSNR", 3), test_utils=full)
    out, stats = ops.get_loss_train(srs, hrs, maps, "cPSNR", 3)
    torch.library.opcheck(ops.get_loss_backward.default, (srs, hrs, maps, stats, torch.rand_like(out), "cPSNR", 3), test_utils=basic)
    # fused Adam (mutates its flat buffers)
    n = 4096
    pbuf, g, m1, v1 = torch.rand(n, device="cuda"), torch.rand(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    torch.library.opcheck(ops.adam_step.default, (pbuf, g, m1, v1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1), test_utils=basic)


def test_forward_is_graph_capturable():
    """The C ABI promises that calls only enqueue work (no allocation, no synchronisation, no host read-back; include/hrnet_hip.h):
    capture HRNet.forward (bf16, all conv kernels incl. the LDS-DMA ones) into a HIP graph and replay it on new inputs."""
    lrs, alphas = synth.fast_batch(61, 2, 6, 64)
    lrs2, _ = synth.fast_batch(62, 2, 6, 64)
    m = util.hip_hrnet("bf16")
    x, a = util.dev(lrs).clone(), util.dev(alphas)
    with torch.no_grad():
        eager1 = m(x, a).clone()                              # warm-up: packs parameters, sizes the workspace, sets LDS limits
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x, a)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = m(x, a)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager1)
        x.copy_(util.dev(lrs2))                               # new data in the captured input buffer
        g.replay()
        torch.cuda.synchronize()
        replayed = out.clone()
        eager2 = m(x, a)
    assert torch.equal(replayed, eager2) and not torch.equal(eager1, eager2)


@pytest.mark.parametrize("S,V", [(18, 2), (27, 3), (50, 4), (100, 2)])
def test_bf16_stages_on_ragged_widths_vs_fp32_path(S, V):
    """Stage by stage (encoder = conv3x3_r64, fusion = conv3x3_v6) at widths that cut lane pairs / quads of the epilogues'
    coalescing exchanges and leave partial tiles: every pixel of the bf16 stage outputs against the exact-fp32 path's, so that a
    mis-addressed piece at an image edge cannot hide in a whole-image PSNR."""
    lrs, alphas = synth.fast_batch(900 + S, 2, V, S)
    x, a = util.dev(lrs), util.dev(alphas)
    m32, m16 = util.hip_hrnet("fp32"), util.hip_hrnet("bf16")
    with torch.no_grad():
        e32, e16 = m32.encode_views(x), m16.encode_views(x)
        assert e16.shape == e32.shape == (2, V, S, S, 64)
        scale = float(e32.abs().max())
        err = (e16.float() - e32.float()).abs()
        assert float(err.max()) <= 3e-2 * scale, (float(err.max()), scale)
        # worst pixel per column must not stand out at the right edge (a wrong piece would be off by O(scale))
        col = err.amax(dim=(0, 1, 2, 4))
        assert float(col[-4:].max()) <= 3.0 * float(col[: max(4, S - 4)].max()) + 1e-3 * scale
        f32, f16 = m32.fuse_views(e32, a), m16.fuse_views(e16, a)
        scale = float(f32.abs().max())
        err = (f16.float() - f32.float()).abs()
        assert float(err.max()) <= 4e-2 * scale, (float(err.max()), scale)
        col = err.amax(dim=(0, 1, 3))
        assert float(col[-4:].max()) <= 3.0 * float(col[: max(4, S - 4)].max()) + 1e-3 * scale


def test_large_image_bf16_vs_fp32_path():
    """BASELINE config 5's image size (512 x 512 -> 1536 x 1536) on a small batch: many tile rows / columns per image, larger in-image
    byte offsets; the bf16 kernels against the exact-fp32 path."""
    lrs, alphas = synth.fast_batch(4242, 1, 3, 512)
    x, a = util.dev(lrs), util.dev(alphas)
    with torch.no_grad():
        ref = util.hip_hrnet("fp32")(x, a).cpu().numpy()
        got = util.hip_hrnet("bf16")(x, a).cpu().numpy()
    assert got.shape == (1, 1, 1536, 1536) and np.isfinite(got).all()
    assert util.rel_err(got, ref) <= BF16_REL and util.psnr_db(got, ref) >= BF16_PSNR
