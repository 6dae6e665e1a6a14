#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  It imports the reference's
own modules (src/DeepNetworks/HRNet.py, src/DeepNetworks/ShiftNet.py, src/lanczos.py and the
pure torch/numpy helpers of src/train.py + src/Evaluator.py), loads the portable seeded
weights of oracle/weights.py into them, evaluates small synthetic cases and stores
inputs + outputs.  Nothing of the reference travels: the fixtures are plain data.

    python oracle/make_goldens.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"
sys.path.insert(0, REF_SRC)

from oracle import synth, weights  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class _Stub(types.ModuleType):
    """Empty stand-in for a plotting / dataset library that is not installed here: any attribute
    resolves to another stub, nothing is ever computed with it."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Stub(self.__name__ + "." + item)

    def __call__(self, *a, **k):
        return None


def _stub_missing_modules():
    """train.py / Evaluator.py / utils.py import plotting + dataset libs that are not installed here;
    their pure tensor helpers do not use them."""
    for name in ["tensorboardX", "skimage", "skimage.io", "seaborn", "sklearn", "sklearn.model_selection",
                 "matplotlib", "matplotlib.pyplot", "mpl_toolkits", "mpl_toolkits.axes_grid1", "pandas"]:
        if name in sys.modules:
            continue
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = _Stub(name)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from DeepNetworks.HRNet import HRNet            # reference
    from DeepNetworks.ShiftNet import ShiftNet      # reference
    import lanczos as ref_lanczos                   # reference

    # ------------------------------------------------------------------ HRNet
    hst = weights.hrnet_state(1234)
    net = HRNet(weights.HRNET_CONFIG).eval()
    net.load_state_dict(weights.to_torch_state(hst))

    def run_hrnet(name, seed, b, v, s, n_real, staged, alpha_residual=True, keep_inputs=True):
        lrs, alphas, hrs = synth.make_batch(seed, b, v, s, n_real)
        net.fuse.alpha_residual = alpha_residual
        with torch.no_grad():
            sr = net(t(lrs), t(alphas)).numpy()
            out = {"sr": sr, "seed": seed, "n_real": np.asarray(n_real if isinstance(n_real, list) else [n_real if n_real is not None else v] * b),
                   "alpha_residual": alpha_residual, "lrs_sum": np.float64(lrs.astype(np.float64).sum())}
            if keep_inputs:
                out.update(lrs=lrs, alphas=alphas)
            else:
                out.update(shape=np.asarray([b, v, s]))
            if staged:
                x = t(lrs).view(-1, v, 1, s, s)
                refs, _ = torch.median(x[:, :9], 1, keepdim=True)
                stacked = torch.cat([x, refs.repeat(1, v, 1, 1, 1)], 2).view(b * v, 2, s, s)
                emb = net.encode(stacked).view(b, v, -1, s, s)
                out["ref"] = refs[:, 0, 0].numpy()
                out["emb"] = emb.numpy()
                fused = net.fuse(emb, t(alphas).view(-1, v, 1, 1, 1))
                out["fused"] = fused.numpy()
        net.fuse.alpha_residual = True
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(f"{name}: sr {sr.shape} |sr|max {np.abs(sr).max():.4f}")

    run_hrnet("hrnet_b2_v5_s16", 11, 2, 5, 16, [5, 3], staged=True)          # odd V, padded views
    run_hrnet("hrnet_b1_v1_s16", 12, 1, 1, 16, None, staged=True)            # V=1: no fusion level
    run_hrnet("hrnet_b1_v12_s24", 13, 1, 12, 24, [9], staged=False)          # median over 9 of 12, pads inside median window? no: 9 real
    run_hrnet("hrnet_b2_v6_s16_pad", 14, 2, 6, 16, [2, 6], staged=True)      # zero views inside the median window
    run_hrnet("hrnet_b2_v4_s16_noalpha", 15, 2, 4, 16, [4, 2], staged=False, alpha_residual=False)
    run_hrnet("hrnet_b1_v32_s32", 16, 1, 32, 32, [27], staged=False)         # 5 fusion levels
    run_hrnet("hrnet_c1_b4_v4_s128", 17, 4, 4, 128, [4, 4, 3, 2], staged=False, keep_inputs=False)  # BASELINE config 1 shape

    # ------------------------------------------------------------------ ShiftNet
    sst = weights.shiftnet_state(4321)
    sn = ShiftNet()
    sn.load_state_dict(weights.to_torch_state(sst))
    rng = np.random.Generator(np.random.PCG64(99))

    def shift_inputs(b):
        lrs, _, _ = synth.make_batch(int(rng.integers(1 << 30)), b, 2, 128)
        x = lrs.copy()                            # (B,2,128,128): pair (reference, view)
        x[:, 1] = np.roll(x[:, 0], (1, 2), axis=(1, 2)) * 0.9 + 0.1 * x[:, 1]
        return x.astype(np.float32)

    def layer_samples(model, x):
        outs = []
        hooks = [getattr(model, f"layer{i}").register_forward_hook(lambda m, i_, o: outs.append(o.detach())) for i in range(1, 9)]
        theta = model(x)
        for h in hooks:
            h.remove()
        samp = {f"layer{i + 1}_sample": o[:, ::8, ::4, ::4].numpy().copy() for i, o in enumerate(outs)}
        samp.update({f"layer{i + 1}_absmean": np.float64(o.abs().double().mean()) for i, o in enumerate(outs)})
        return theta, samp

    sn.eval()
    x = shift_inputs(3)
    with torch.no_grad():
        theta, samp = layer_samples(sn, t(x))
    np.savez_compressed(os.path.join(OUT, "shiftnet_eval_b3.npz"), x=x, theta=theta.numpy(), **samp)
    print("shiftnet eval theta", theta.numpy())

    sn.train()
    x = shift_inputs(4)
    cap = {}
    hk = sn.drop1.register_forward_hook(lambda m, i_, o: cap.update(inp=i_[0].detach(), out=o.detach()))
    torch.manual_seed(5)
    with torch.no_grad():
        theta, samp = layer_samples(sn, t(x))
    hk.remove()
    mask = (cap["out"] != 0).numpy().astype(np.uint8)        # where the input is 0 the mask value is irrelevant
    run_mean = {f"layer{i}_running_mean": getattr(sn, f"layer{i}")[1].running_mean.numpy().copy() for i in range(1, 9)}
    run_var = {f"layer{i}_running_var": getattr(sn, f"layer{i}")[1].running_var.numpy().copy() for i in range(1, 9)}
    np.savez_compressed(os.path.join(OUT, "shiftnet_train_b4.npz"), x=x, theta=theta.numpy(), dropout_mask=np.packbits(mask, axis=1),
                        **samp, **run_mean, **run_var)
    print("shiftnet train theta", theta.numpy())
    sn.load_state_dict(weights.to_torch_state(sst))          # restore running stats
    sn.eval()

    # ------------------------------------------------------------------ Lanczos
    d = np.array([0.0, 0.3, -0.7, 1.0, 2.5, -3.2, 1e-7, 0.5, -0.49999], np.float32).reshape(-1, 1)
    taps = ref_lanczos.lanczos_kernel(t(d), a=3, N=7).numpy()
    img = (rng.random((2, 4, 40, 36), dtype=np.float32)).astype(np.float32)
    shift = np.array([[0.0, 0.0], [0.37, -0.81], [-1.6, 2.2], [1.0, -1.0]], np.float32)
    shifted = ref_lanczos.lanczos_shift(t(img), t(shift), p=3, a=3, N=7).numpy()
    theta = np.array([[0.25, -0.5], [0.0, 0.0], [-1.3, 0.9], [0.6, 0.6], [2.0, -2.75]], np.float32)
    imgs = rng.random((5, 1, 48, 48), dtype=np.float32)
    with torch.no_grad():
        tr = sn.transform(t(theta), t(imgs)).numpy()
    np.savez_compressed(os.path.join(OUT, "lanczos.npz"), d=d, taps=taps, img=img, shift=shift, shifted=shifted,
                        theta=theta, imgs=imgs, transformed=tr)
    print("lanczos taps row1", taps[1], "transform", tr.shape)

    # ------------------------------------------------------------------ callers (train.py / Evaluator.py helpers)
    _stub_missing_modules()
    try:
        import train as ref_train
        import Evaluator as ref_eval
        b, s = 3, 32
        lrs, alphas, hrs = synth.make_batch(21, b, 4, s)
        with torch.no_grad():
            srs = net(t(lrs), t(alphas))[:, 0]
        maps = (rng.random((b, 3 * s, 3 * s)) > 0.1).astype(np.float32)
        crop = ref_train.get_crop_mask(s, 3).numpy()
        l_cpsnr = ref_train.get_loss(srs, t(hrs), t(maps) * t(crop)[0], metric="cPSNR").numpy()
        l_cmse = ref_train.get_loss(srs, t(hrs), t(maps) * t(crop)[0], metric="cMSE").numpy()
        srn = np.clip(srs.numpy(), 0, 1)
        sc = np.array([ref_eval.shift_cPSNR(srn[i].astype(np.float64), hrs[i].astype(np.float64), maps[i].astype(np.float64)) for i in range(b)])
        cp = np.array([ref_eval.cPSNR(srn[i].astype(np.float64), hrs[i].astype(np.float64), maps[i].astype(np.float64)) for i in range(b)])
        # registration glue on a 128x128 SR crop (train.py:26-63): thetas from the reference ShiftNet, shifted SRs
        lrs2, alphas2, hrs2 = synth.make_batch(22, 2, 2, 48)          # SR 144x144 -> centre crop 128
        with torch.no_grad():
            srs2 = net(t(lrs2), t(alphas2))
            off = (144 - 128) // 2
            shifts = ref_train.register_batch(sn, srs2[:, :, off:off + 128, off:off + 128],
                                              reference=t(hrs2)[:, off:off + 128, off:off + 128].view(-1, 1, 128, 128))
            shifted2 = ref_train.apply_shifts(sn, srs2, shifts, "cpu")[:, 0]
        np.savez_compressed(os.path.join(OUT, "callers.npz"), srs=srs.numpy(), hrs=hrs, maps=maps, crop=crop,
                            loss_cpsnr=l_cpsnr, loss_cmse=l_cmse, shift_cpsnr=sc, cpsnr=cp,
                            lrs2=lrs2, alphas2=alphas2, hrs2=hrs2, srs2=srs2.numpy(), shifts=shifts.numpy(), shifted2=shifted2.numpy())
        print("callers: loss", l_cpsnr, "shift_cPSNR", sc, "shifts", shifts.numpy().ravel())
    except Exception as e:  # ordinary import error of an optional helper: record and continue
        print("callers golden skipped:", repr(e))

    # ------------------------------------------------------------------ one full train step of the reference (train.py:164-190), fp64
    train_step_golden(HRNet, ShiftNet)


def train_step_golden(HRNet, ShiftNet):
    """srs = fusion_model(lrs, alphas); shifts = register_batch(...); srs_shifted = apply_shifts(...); loss = -get_loss(..., 'cPSNR')
    mean + lambda mean(shifts)^2; loss.backward() - the reference's own statements and modules, in fp64, train mode (BatchNorm batch
    statistics; the dropout mask is injected through a forward hook so that the GPU test can feed the same one).  Stored: loss,
    shifts, a crop of the SR image, and per parameter the gradient's L2 norm, sum and a strided sample (every tensor is far too
    large to keep whole: fc1.weight alone is 33.5 M values); for the nine single-slope PReLUs and decode.final.bias additionally
    sum |terms| of the gradient's defining sum (captured with module hooks), which is what bounds a float32 implementation's
    error on those cancelling sums.  Inputs are regenerated from their seeds by the test."""
    import train as ref_train
    B, V, S, lam, crop_w = 2, 3, 48, 1e-6, 3
    lrs, alphas, hrs = synth.make_batch(31, B, V, S, V)
    rng = np.random.Generator(np.random.PCG64(5))
    maps = (rng.random((B, 3 * S, 3 * S)) > 0.1).astype(np.float32)
    keep = (rng.random((B, 32768)) >= 0.5)                                   # dropout keep-mask, reference flatten order
    off = (3 * S - 128) // 2

    fusion = HRNet(weights.HRNET_CONFIG).double().train()
    fusion.load_state_dict({k: v.double() for k, v in weights.to_torch_state(weights.hrnet_state(1234)).items()})
    regis = ShiftNet().double().train()
    regis.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in weights.to_torch_state(weights.shiftnet_state(4321)).items()})
    tmask = t(keep.astype(np.float64))
    hk = regis.drop1.register_forward_hook(lambda m, i_, o: i_[0] * tmask / 0.5)      # dropout(p=0.5) with OUR mask

    # sum |terms| of the scalar parameters' gradients: PReLU slope a: d a = sum g * min(x, 0);  decode.final.bias: sum g
    abs_terms, hooks, saved = {}, [], {}
    for name, mod in fusion.named_modules():
        if isinstance(mod, torch.nn.PReLU):
            hooks.append(mod.register_forward_hook(lambda m, i_, o, name=name: saved.__setitem__(name, i_[0].detach())))
            hooks.append(mod.register_full_backward_hook(
                lambda m, gi, go, name=name: abs_terms.__setitem__(name + ".weight", float((go[0] * saved[name].clamp(max=0)).abs().sum()))))
    hooks.append(fusion.decode.final.register_full_backward_hook(
        lambda m, gi, go: abs_terms.__setitem__("decode.final.bias", float(go[0].abs().sum()))))

    t_lrs, t_alphas, t_hrs, t_maps = t(lrs).double(), t(alphas).double(), t(hrs).double(), t(maps).double()
    torch_mask = ref_train.get_crop_mask(patch_size=S, crop_size=crop_w).double()
    srs = fusion(t_lrs, t_alphas)
    shifts = ref_train.register_batch(regis, srs[:, :, off:off + 128, off:off + 128],
                                      reference=t_hrs[:, off:off + 128, off:off + 128].view(-1, 1, 128, 128))
    srs_shifted = ref_train.apply_shifts(regis, srs, shifts, "cpu")[:, 0]
    cropped_mask = torch_mask[0] * t_maps
    loss = -ref_train.get_loss(srs_shifted, t_hrs, cropped_mask, metric="cPSNR")
    loss = torch.mean(loss)
    loss = loss + lam * torch.mean(shifts) ** 2
    loss.backward()
    hk.remove()
    for h in hooks:
        h.remove()

    out = {"shape": np.asarray([B, V, S]), "lam": lam, "crop": crop_w, "loss": float(loss.detach()), "shifts": shifts.detach().numpy(),
           "srs_crop": srs.detach().numpy()[:, :, 40:72, 40:72], "srs_shifted_crop": srs_shifted.detach().numpy()[:, 40:72, 40:72]}
    for prefix, model in (("hrnet", fusion), ("shiftnet", regis)):
        for k, p in model.named_parameters():
            g = p.grad.detach().numpy().ravel()
            stride = max(1, g.size // 2048)
            out[f"{prefix}/{k}/norm"] = float(np.sqrt((g * g).sum()))
            out[f"{prefix}/{k}/sum"] = float(g.sum())
            out[f"{prefix}/{k}/absmax"] = float(np.abs(g).max())
            out[f"{prefix}/{k}/stride"] = stride
            out[f"{prefix}/{k}/sample"] = g[::stride].copy()
            if prefix == "hrnet" and k in abs_terms:
                out[f"{prefix}/{k}/abs_terms"] = abs_terms[k]
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)
    print("train_step: loss", out["loss"], "shifts", shifts.detach().numpy().ravel(), "scalars with abs_terms:", sorted(abs_terms))


if __name__ == "__main__":
    main()
