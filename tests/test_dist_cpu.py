"""CPU, world_size 2, gloo: the N>1 plumbing bench.py relies on (rendezvous on 127.0.0.1, barrier, max-over-ranks
timing, contiguous batch shards, whole-job frame count)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, os.path.join(%r, "highres-net_amd"))
    import torch
    from hrnet_hip import dist as hdist
    rank, local_rank, ws = hdist.init(backend="gloo")
    assert ws == 2 and rank in (0, 1)
    lo, hi = hdist.shard(64, rank, ws)
    assert (lo, hi) == (rank * 32, rank * 32 + 32)
    hdist.barrier()
    elapsed = 1.0 + rank            # rank 1 is the slow one
    t = hdist.max_over_ranks(elapsed)
    frames = hdist.sum_over_ranks(hi - lo)
    assert t == 2.0 and frames == 64.0, (t, frames)
    try:
        hdist.shard(33, rank, ws)
        raise SystemExit("shard must reject a ragged batch")
    except ValueError:
        pass
    hdist.barrier()
    if rank == 0:
        print(json.dumps({"value": frames / t, "n_gpus": ws}))
    hdist.finalize()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_plumbing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert '"value": 32.0' in outs[0][0]


GRAD_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "highres-net_amd"))
    import torch
    from hrnet_hip import dist as hdist
    rank, local_rank, ws = hdist.init(backend="gloo")
    torch.manual_seed(0)                                  # identical parameters on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.PReLU(), torch.nn.Linear(200, 7))
    other = torch.nn.Linear(5, 5)                         # a second module; its bias gets no gradient on rank 1
    x = torch.full((4, 300), float(rank + 1))
    (net(x).sum() + (other.weight.sum() if rank == 1 else other(torch.ones(1, 5)).sum())).backward()
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for m in (net, other) for p in m.parameters()]
    nbytes = hdist.allreduce_gradients([net, other], bucket_mb=0.1)       # 0.1 MiB buckets: several buckets for 0.25 MB
    assert nbytes == sum(g.numel() * 4 for g in local), nbytes
    # reference: gather every rank's local gradients and average them by hand
    for g, p in zip(local, [p for m in (net, other) for p in m.parameters()]):
        both = [torch.zeros_like(g) for _ in range(ws)]
        torch.distributed.all_gather(both, g)
        want = sum(both) / ws
        assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), (p.shape, (p.grad - want).abs().max())
    hdist.barrier()
    if rank == 0:
        print("grads averaged")
    hdist.finalize()
""") % ROOT


def test_two_rank_gradient_allreduce(tmp_path):
    """The one exchange step of data-parallel training: bucketed average of the gradients, here on gloo."""
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "grads averaged" in outs[0][0]


TRAIN_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "highres-net_amd"))
    import torch
    from hrnet_hip import dist as hdist
    rank, local_rank, ws = hdist.init(backend="gloo")
    torch.manual_seed(0)                                  # identical parameters on both ranks
    # CPU stand-ins for the two models of train.py: `fusion` (runs first in forward, LAST in backward) and `regis` (its gradients
    # are complete first), re-homed into one flat buffer exactly as hrnet_hip.optim.FusedAdam does it
    fusion = torch.nn.Sequential(torch.nn.Linear(64, 48), torch.nn.PReLU(), torch.nn.Linear(48, 32))
    regis = torch.nn.Sequential(torch.nn.Linear(32, 400), torch.nn.ReLU(), torch.nn.Linear(400, 2, bias=False))
    params = list(fusion.parameters()) + list(regis.parameters())
    n = sum(p.numel() for p in params)
    flat_p, flat_g = torch.zeros(n), torch.zeros(n)
    off = 0
    with torch.no_grad():
        for p in params:
            k = p.numel()
            flat_p[off:off + k].copy_(p.reshape(-1)); p.data = flat_p[off:off + k].view(p.shape); p.grad = flat_g[off:off + k].view(p.shape)
            off += k
    buckets = hdist.GradBuckets(flat_g, params, early=list(regis.parameters()))
    opt = torch.optim.SGD(params, lr=0.05)
    seen = []
    fusion[0].weight.register_hook(lambda g: seen.append(buckets.early_launched_in_backward))   # fires in the LAST backward node
    for step in range(3):
        flat_g.zero_(); buckets.begin()
        x = torch.randn(8, 64, generator=torch.Generator().manual_seed(100 * rank + step))     # every rank its own shard
        loss = (regis(fusion(x)) ** 2).mean()
        loss.backward()
        local = flat_g.clone()
        nbytes = buckets.finish()
        assert nbytes == n * 4
        both = [torch.zeros_like(local) for _ in range(ws)]
        torch.distributed.all_gather(both, local)
        assert torch.allclose(flat_g, sum(both) / ws, rtol=1e-6, atol=1e-8)
        opt.step()
    assert seen == [True] * 3, seen                        # the early slice was on the wire before fusion's backward finished
    # a step after `module.zero_grad()` (set_to_none=True, torch's default): autograd installs FRESH .grad tensors, the flat slice
    # holds nothing of this step - the hook must not put it on the wire; the optimiser folds the gradients back in (FusedAdam._rebind)
    # and finish() reduces everything afterwards
    flat_g.zero_(); buckets.begin()
    regis.zero_grad()
    assert all(p.grad is None for p in regis.parameters())
    x = torch.randn(8, 64, generator=torch.Generator().manual_seed(100 * rank + 9))
    (regis(fusion(x)) ** 2).mean().backward()
    assert not buckets.early_launched_in_backward
    off = 0
    for p in params:
        k = p.numel()
        view = flat_g[off:off + k].view(p.shape)
        if p.grad.data_ptr() != view.data_ptr():
            view.add_(p.grad); p.grad = view
        off += k
    local = flat_g.clone()
    assert float(local[buckets.lo:buckets.hi].abs().sum()) > 0
    buckets.finish()
    both = [torch.zeros_like(local) for _ in range(ws)]
    torch.distributed.all_gather(both, local)
    assert torch.allclose(flat_g, sum(both) / ws, rtol=1e-6, atol=1e-8)
    try:
        buckets.finish()
        raise SystemExit("a second finish() in one step must be refused")
    except RuntimeError:
        pass
    # a rebuilt bucket takes over: the old one's hooks are gone once it is closed
    buckets.close()
    b2 = hdist.GradBuckets(flat_g, params, early=list(regis.parameters()))
    flat_g.zero_(); b2.begin()
    (regis(fusion(x)) ** 2).mean().backward()
    assert b2.early_launched_in_backward and not buckets.early_launched_in_backward
    b2.finish()
    mine = flat_p.clone()
    both = [torch.zeros_like(mine) for _ in range(ws)]
    torch.distributed.all_gather(both, mine)
    assert torch.equal(both[0], both[1])                   # rank-identical parameters after the steps
    hdist.barrier()
    if rank == 0:
        print("train loop ok")
    hdist.finalize()
""") % ROOT


def test_two_rank_train_loop_with_overlapped_exchange(tmp_path):
    """BASELINE configs[3] (data-parallel `src/train.py` loop) on CPU stand-ins: the early bucket (the registration model's
    gradients) is all-reduced from a backward hook while the fusion model's backward still runs, the rest after backward; the
    averaged gradients equal the hand-gathered mean and both ranks hold identical parameters after three optimiser steps."""
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "train loop ok" in outs[0][0]


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` launched directly (no torchrun environment) starts two ranks itself, before any GPU call, and rank
    0's line says n_gpus 2 with both ranks seen by an all_gather.  Here: no ROCm device, HRN_DIST_BACKEND=gloo = the launcher
    rehearsal on CPU stand-in steps (no kernel runs, value is null).  Without that variable, or with fewer devices than ranks, the
    command refuses instead of silently measuring one GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for mode in ("forward", "train"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode],
                           env=dict(env, HRN_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["backend"] == "gloo"
        assert line["value"] is None and line["rehearsal"] is True and line["mode"] == mode
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


VAL_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "highres-net_amd"))
    sys.path.insert(0, %r)
    import numpy as np, torch
    from hrnet_hip import dist as hdist, validate
    from oracle import hrnet_np as O
    rank, local_rank, ws = hdist.init(backend="gloo")
    # CPU stand-ins: a "fusion model" (bicubic x3 of the first view) and the numpy oracle's shift_cPSNR as the scorer; 7 imagesets of
    # batch size 1 (train.py:281), dealt round-robin: rank 0 scores 4, rank 1 scores 3
    class Toy(torch.nn.Module):
        def forward(self, lrs, alphas):
            return torch.nn.functional.interpolate(lrs[:, :1], scale_factor=3, mode="bicubic", align_corners=False)
    def score(srs, hrs, maps):
        return torch.tensor([O.shift_cpsnr(np.clip(s.numpy(), 0, 1), h.numpy(), m.numpy()) for s, h, m in zip(srs, hrs, maps)])
    g = torch.Generator().manual_seed(3)
    sets = []
    for i in range(7):
        lrs = torch.rand(1, 3, 16, 16, generator=g)
        sets.append((lrs, torch.ones(1, 3), torch.rand(1, 48, 48, generator=g), (torch.rand(1, 48, 48, generator=g) > 0.1).float()))
    model = Toy().train()
    mine = [sets[i] for i in validate.shard_indices(len(sets), rank, ws)]
    got = validate.sharded_val_score(model, mine, score_fn=score)
    assert model.training                                   # the caller's mode is restored (train.py:196 / :160)
    want = -float(np.mean([float(score(model(l, a)[:, 0], h, m)[0]) for l, a, h, m in sets]))
    assert abs(got - want) <= 1e-12 * abs(want), (got, want)
    hdist.barrier()
    if rank == 0:
        print("val score ok", got)
    hdist.finalize()
""") % (ROOT, ROOT)


def test_two_rank_sharded_validation(tmp_path):
    """train.py:196-215 sharded by imageset: every rank scores its own imagesets, one all-reduce of (sum, count); the
    reduced val_score equals the single-process one.  CPU stand-ins for the model and for hrn_shift_cpsnr (the numpy oracle's)."""
    script = tmp_path / "val_worker.py"
    script.write_text(VAL_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "val score ok" in outs[0][0]
