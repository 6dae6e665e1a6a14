"""GPU (-m gpu): the loss / score reductions behind the hot path, through the C ABI,
against the values the reference's own `train.get_loss`, `Evaluator.cPSNR` and `Evaluator.shift_cPSNR` produced
(tests/golden/callers.npz, written by oracle/make_goldens.py) and against the numpy oracle on a full-size batch."""
import numpy as np
import pytest
import torch

from oracle import hrnet_np as O
import util

pytestmark = pytest.mark.gpu


def test_get_loss_vs_reference_golden():
    from hrnet_hip import binding
    g = util.golden("callers")
    srs, hrs = util.dev(g["srs"]), util.dev(g["hrs"])
    mask = g["maps"] * g["crop"][0]                     # train.py:183: torch_mask[0] * hr_maps
    for metric, key in (("cPSNR", "loss_cpsnr"), ("cMSE", "loss_cmse")):
        got = binding.get_loss(srs, hrs, util.dev(mask.astype(np.float32)), metric).cpu().numpy()
        assert util.rel_err(got, g[key]) <= 1e-5
        # get_crop_mask folded into the kernel: same result from the raw status maps + crop=3
        got2 = binding.get_loss(srs, hrs, util.dev(g["maps"]), metric, crop=3).cpu().numpy()
        assert util.rel_err(got2, g[key]) <= 1e-5
    mm = binding.get_loss(srs, hrs, util.dev(mask.astype(np.float32)), "masked_MSE").cpu().numpy()
    want = ((mask * g["srs"] - mask * g["hrs"]) ** 2).mean(axis=(1, 2))
    assert util.rel_err(mm, want) <= 1e-5
    with pytest.raises(ValueError):
        binding.get_loss(srs, hrs, util.dev(g["maps"]), "PSNR")


def test_shift_cpsnr_vs_reference_golden():
    from hrnet_hip import binding
    g = util.golden("callers")
    got = binding.shift_cpsnr(util.dev(g["srs"]), util.dev(g["hrs"]), util.dev(g["maps"]), border_w=3, clip=True).cpu().numpy()
    assert util.rel_err(got, g["shift_cpsnr"]) <= 1e-5
    # border 0 == plain cPSNR of the clipped image
    got0 = binding.shift_cpsnr(util.dev(g["srs"]), util.dev(g["hrs"]), util.dev(g["maps"]), border_w=0, clip=True).cpu().numpy()
    assert util.rel_err(got0, g["cpsnr"]) <= 1e-5


def test_scores_full_size_vs_oracle():
    """B=32 frames of 384x384 (the metric's output size): device reductions vs the fp64 numpy oracle."""
    from hrnet_hip import binding
    rng = np.random.Generator(np.random.PCG64(4))
    hr = rng.random((32, 384, 384), dtype=np.float32) * 0.25
    sr = np.clip(hr + 0.01 * rng.standard_normal(hr.shape).astype(np.float32) + 0.003, 0, 1).astype(np.float32)
    sr = np.roll(sr, (1, -2), axis=(1, 2))              # a registration error the shift search should undo
    mp = (rng.random(hr.shape) > 0.05).astype(np.float32)
    got = binding.shift_cpsnr(util.dev(sr), util.dev(hr), util.dev(mp)).cpu().numpy()
    want = O.shift_cpsnr(sr.astype(np.float64), hr.astype(np.float64), mp.astype(np.float64))
    assert util.rel_err(got, want) <= 1e-5
    plain = binding.shift_cpsnr(util.dev(sr), util.dev(hr), util.dev(mp), border_w=0).cpu().numpy()
    assert (got > plain + 3.0).all()                    # the best shift beats the unregistered score by a wide margin
    loss = binding.get_loss(util.dev(sr), util.dev(hr), util.dev(mp), "cPSNR").cpu().numpy()
    assert util.rel_err(loss, O.get_loss(sr, hr, mp, "cPSNR")) <= 1e-5


def test_hrnet_large_tiles_512():
    """BASELINE config 5 geometry (512x512 -> 1536x1536 tiles; small batch here): the persistent tile walk, halo logic
    and the staging buffers at 16x the pixels per image.  Properties: determinism, batch independence, and agreement of
    the bf16 path with the exact-fp32 path within the bf16 tolerance."""
    from oracle import synth
    lrs, alphas = synth.fast_batch(11, 2, 6, 512)
    x, a = util.dev(lrs), util.dev(alphas)
    m16, m32 = util.hip_hrnet("bf16"), util.hip_hrnet("fp32")
    with torch.no_grad():
        y = m16(x, a).clone()
        assert y.shape == (2, 1, 1536, 1536) and torch.isfinite(y).all()
        assert torch.equal(m16(x, a), y)
        assert torch.equal(m16(x[1:2].contiguous(), a[1:2].contiguous())[0], y[1])
        ref = m32(x, a)
    assert util.rel_err(y.cpu().numpy(), ref.cpu().numpy()) <= 4e-2
    assert util.psnr_db(y.cpu().numpy(), ref.cpu().numpy()) >= 42.0


def test_sharded_val_score_single_rank_on_device():
    """hrnet_hip.validate.sharded_val_score without a process group = the reference's validation loop (train.py:196-215) on device:
    HRNet in eval mode, hrn_shift_cpsnr (clip to [0, 1], 7 x 7 offsets) per sample, minus the mean; the model's mode is restored."""
    from hrnet_hip import validate
    from oracle import synth
    import util
    m = util.hip_hrnet("fp32").train()
    sets = []
    for i in range(3):
        lrs, alphas, hrs = synth.make_batch(40 + i, 1, 4, 32, 4)
        maps = (np.random.Generator(np.random.PCG64(i)).random((1, 96, 96)) > 0.1).astype(np.float32)
        sets.append((util.dev(lrs), util.dev(alphas), util.dev(hrs), util.dev(maps)))
    got = validate.sharded_val_score(m, sets)
    assert m.training
    m.eval()
    want = []
    with torch.no_grad():
        for lrs, alphas, hrs, maps in sets:
            sr = m(lrs, alphas)[:, 0].clamp(0, 1).cpu().numpy()
            want.append(O.shift_cpsnr(sr[0], hrs.cpu().numpy()[0], maps.cpu().numpy()[0]))
    assert abs(got + float(np.mean(want))) <= 1e-4 * abs(float(np.mean(want))), (got, want)
