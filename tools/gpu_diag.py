#!/usr/bin/env python3
"""GPU-box diagnostic: runs every HIP stage against the goldens / oracle and PRINTS the errors (no asserts).
Usage on the box:  python tools/gpu_diag.py > gpurun_out/diag.log 2>&1"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "highres-net_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import hrnet_np as O  # noqa: E402
from oracle import synth, weights  # noqa: E402
import util  # noqa: E402


def section(name):
    print(f"\n===== {name}", flush=True)


def guarded(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()
    sys.stdout.flush()


def hrnet_cases():
    hst = weights.hrnet_state(1234)
    for prec in ("fp32", "bf16"):
        for name in ["hrnet_b1_v1_s16", "hrnet_b2_v5_s16", "hrnet_b2_v6_s16_pad", "hrnet_b1_v12_s24", "hrnet_b2_v4_s16_noalpha", "hrnet_b1_v32_s32"]:
            g = util.golden(name)
            m = util.hip_hrnet(prec, bool(g["alpha_residual"]))
            lrs, alphas = util.dev(g["lrs"]), util.dev(g["alphas"])
            with torch.no_grad():
                line = f"{prec} {name}:"
                if "emb" in g.files:
                    ref = O.reference_frame(g["lrs"].astype(np.float64))
                    emb = m.encode_views(lrs)
                    e = util.nhwc_to_nchw(emb)
                    line += f" emb {util.rel_err(e, g['emb']):.2e}"
                    fused = m.fuse_views(emb.clone(), alphas)
                    f = util.nhwc_to_nchw(fused)
                    line += f" fused {util.rel_err(f, g['fused']):.2e}"
                    # decoder alone on the golden fused state
                    gf = torch.from_numpy(g["fused"]).cuda().permute(0, 2, 3, 1).contiguous().to(fused.dtype)
                    line += f" dec {util.rel_err(m.decode_state(gf).cpu().numpy(), g['sr']):.2e}"
                sr = m(lrs, alphas).cpu().numpy()
                line += f" sr {util.rel_err(sr, g['sr']):.2e} psnr {util.psnr_db(sr, g['sr']):.1f} dB"
            print(line, flush=True)


def c1_case():
    g = util.golden("hrnet_c1_b4_v4_s128")
    b, v, s = (int(x) for x in g["shape"])
    lrs, alphas, _ = synth.make_batch(int(g["seed"]), b, v, s, [int(x) for x in g["n_real"]])
    for prec in ("fp32", "bf16"):
        m = util.hip_hrnet(prec)
        with torch.no_grad():
            sr = m(util.dev(lrs), util.dev(alphas))
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                sr = m(util.dev(lrs), util.dev(alphas))
            torch.cuda.synchronize()
            dt = (time.time() - t0) / 3
        sr = sr.cpu().numpy()
        print(f"c1 {prec}: sr {util.rel_err(sr, g['sr']):.2e} psnr {util.psnr_db(sr, g['sr']):.1f} dB  {dt * 1e3:.2f} ms/fwd", flush=True)


def shiftnet_cases():
    g = util.golden("shiftnet_eval_b3")
    m = util.hip_shiftnet()
    with torch.no_grad():
        th = m(util.dev(g["x"])).cpu().numpy()
    print("shiftnet eval theta\n", th, "\nref\n", g["theta"], "\nrel", util.rel_err(th, g["theta"]), flush=True)
    g = util.golden("shiftnet_train_b4")
    m = util.hip_shiftnet().train()
    mask = np.unpackbits(g["dropout_mask"], axis=1)[:, :32768]
    from hrnet_hip import binding
    with torch.no_grad():
        th = binding.shiftnet_forward(m.packed_parameters(), m._named(), util.dev(g["x"]), train_bn=True, momentum=0.1,
                                      dropout_mask=util.dev(mask.astype(np.uint8))).cpu().numpy()
    print("shiftnet train theta\n", th, "\nref\n", g["theta"], "\nrel", util.rel_err(th, g["theta"]), flush=True)
    for i in (1, 5, 8):
        rm = getattr(m, f"layer{i}")[1].running_mean.cpu().numpy()
        rv = getattr(m, f"layer{i}")[1].running_var.cpu().numpy()
        print(f"  layer{i} running_mean err {np.abs(rm - g[f'layer{i}_running_mean']).max():.2e} running_var err {np.abs(rv - g[f'layer{i}_running_var']).max():.2e}")


def lanczos_cases():
    import lanczos
    g = util.golden("lanczos")
    taps = lanczos.lanczos_kernel(util.dev(g["d"])).cpu().numpy()
    print("taps err", np.abs(taps - g["taps"]).max())
    out = lanczos.lanczos_shift(util.dev(g["img"]), util.dev(g["shift"]), p=3).cpu().numpy()
    print("shift err", np.abs(out - g["shifted"]).max())
    m = util.hip_shiftnet()
    tr = m.transform(util.dev(g["theta"]), util.dev(g["imgs"])).cpu().numpy()
    print("transform", tr.shape, "err", np.abs(tr - g["transformed"]).max(), flush=True)


def timing():
    for prec, b, v in (("bf16", 8, 8), ("bf16", 32, 32), ("fp32", 16, 16)):
        lrs, alphas = synth.fast_batch(7, b, v, 128)
        m = util.hip_hrnet(prec)
        x, a = util.dev(lrs), util.dev(alphas)
        with torch.no_grad():
            m(x, a)
            torch.cuda.synchronize()
            t0 = time.time()
            n = 3
            for _ in range(n):
                m(x, a)
            torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        gf = b * (6.078 * v + 12.080 * (v - 1) + 1.227)
        print(f"timing {prec} B={b} V={v}: {dt * 1e3:.2f} ms/fwd  {b / dt:.1f} frames/s  {gf / dt / 1e3:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.version.hip)
    for name, fn in (("hrnet small", hrnet_cases), ("hrnet c1", c1_case), ("lanczos", lanczos_cases),
                     ("shiftnet", shiftnet_cases), ("timing", timing)):
        section(name)
        guarded(fn)
