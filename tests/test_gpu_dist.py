"""GPU, world_size 2 on ONE card: the data-parallel train step (BASELINE configs[3]/[4]) on the real HIP backward pass.

Two ranks share cuda:0 and exchange over gloo (RCCL refuses two ranks on one device; gloo moves device tensors through the
host, which is all this test needs): each rank runs bench.py's train step on ITS OWN synthetic batch, so the local gradients
differ, and after two optimisation steps every parameter must still be bit-identical on both ranks - which holds only if
every slice of the flat gradient buffer was averaged.  Also checked: ShiftNet's slice went on the wire from the backward hook
(before `allreduce()` was called), and a step with the exchange switched off leaves the ranks apart."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from test_dist_cpu import _free_port, ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "highres-net_amd"))
    import torch
    import torch.distributed as dist
    from hrnet_hip import dist as hdist
    rank, local_rank, ws = hdist.init(backend="gloo")
    import bench
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    step = bench.make_train_step(device, 4, 8, 64)
    flat = step.flat_parameters

    def digests():
        torch.cuda.synchronize(device)
        mine = flat.detach().cpu()
        both = [None, None]
        dist.all_gather_object(both, mine)
        return both

    a, b = digests()
    assert torch.equal(a, b), "the ranks must start from the same weights"
    losses = []
    for _ in range(2):
        losses.append(float(step()))
        assert step.early, "ShiftNet's slice must be on the wire before allreduce() is called"
    a, b = digests()
    same = bool(torch.equal(a, b))
    moved = float((a - step.initial.cpu()).abs().max())
    step(exchange=False)
    a, b = digests()
    apart = float((a - b).abs().max())
    both_losses = [None, None]
    dist.all_gather_object(both_losses, losses)
    if rank == 0:
        print(json.dumps({"same": same, "moved": moved, "apart": apart, "losses": both_losses}))
    hdist.finalize()
""") % (ROOT, ROOT)


def test_two_ranks_one_card_train_steps_stay_in_lockstep(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["same"], "parameters diverged: some slice of the gradient buffer was not averaged"
    assert res["moved"] > 1e-6, "the optimiser did not move the weights"
    assert res["apart"] > 1e-7, "different batches must give different updates once the exchange is off"
    assert res["losses"][0] != res["losses"][1], "the ranks were meant to see different batches"
