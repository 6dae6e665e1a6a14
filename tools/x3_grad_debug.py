"""Per-parameter gradient error of the bf16x3 training mode against the fp64 autograd oracle (and the fp32 HIP path), for one shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "highres-net_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import conftest  # noqa
import util, test_gpu_backward as T
from oracle import synth
B, V, S = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 4, 16)))
lrs, alphas, _ = synth.make_batch(5, B, V, S, V)
rng = np.random.Generator(np.random.PCG64(77))
cot = rng.standard_normal((B, 1, 3 * S, 3 * S)).astype(np.float32)
SL = None
if os.environ.get("SLOPE"):
    from oracle import weights
    SL = {k: float(os.environ["SLOPE"]) for k in weights.hrnet_state(1234) if k.endswith((".1.weight", ".3.weight", "fuse.2.weight")) and "block" in k or k in ("encode.init_layer.1.weight", "fuse.fuse.2.weight", "decode.deconv.1.weight")}
want_sr, want = T._oracle_grads(lrs, alphas, cot, True, slopes=SL)
res = {}
for prec in ("fp32", "bf16x3"):
    m = T._fresh_model(True, precision=prec, slopes=SL)
    sr = m(util.dev(lrs), util.dev(alphas))
    (sr * util.dev(cot)).sum().backward()
    res[prec] = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
    print(prec, "sr err", util.rel_err(sr.detach().cpu().numpy(), want_sr))
for k in res["fp32"]:
    if res["fp32"][k].size > 1:
        print(f"{k:45s} fp32 {util.rel_err(res['fp32'][k], want[k]):.2e}  x3 {util.rel_err(res['bf16x3'][k], want[k]):.2e}")
