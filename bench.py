#!/usr/bin/env python3
"""bench.py - SR frames/s of the MI355X HighRes-net forward (BASELINE.json metric) + parity + rooflines + CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE HRNet.forward(lrs, alphas) on one resident batch of the workload the metric is quoted on
(BASELINE.json configs[2]: B=32, n_views=32, 128x128 -> 384x384, bf16 storage / fp32 accumulation).  With N GPUs
every rank runs its own batch (weak scaling, no data-path collective: SURVEY.md section 8e); `value` = N*B*K / max-over-ranks
wall time.  Rank 0 prints one JSON line with the contract keys plus
  roofline      dominant kernel family (MFMA bound), measured live with the library's hipEvent profiler over extra steps
  roofline_hbm  the recursive-fusion stage against the HBM roof (north_star): SURVEY 8d's algorithmic bytes / its measured time
  parity        the metric's second half ("cPSNR vs reference"): the timed bf16 output against the exact-fp32 HIP path on the same
                batch, and (N=1) both HIP paths against the torch-CPU port on the cpu_baseline sample
  fp32_path / c2_fp32 / train_step   extra measurements: the exact-fp32 path at the metric's shape, BASELINE configs[1]
                (B=16, V=16, fp32), and one optimisation step of src/train.py at the reference's training shape
  cpu_baseline  own torch-CPU port of the same forward on a bounded sample of the same workload, host cores stated (N=1 only).

    python bench.py --mode train [--gpus N] ...   BASELINE configs[3]: the data-parallel optimisation step of src/train.py:164-191
on the HIP modules (per-rank B=32, 32 views, 64x64 patches, fp32), FusedAdam, gradient exchange over RCCL with ShiftNet's 137 MB
slice all-reduced from a backward hook under HRNet's backward; value = samples/s over all ranks.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "highres-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

NETWORK = {   # reference config/config.json:8-34
    "encoder": {"in_channels": 2, "num_layers": 2, "kernel_size": 3, "channel_size": 64},
    "recursive": {"alpha_residual": True, "in_channels": 64, "num_layers": 2, "kernel_size": 3},
    "decoder": {"deconv": {"in_channels": 64, "kernel_size": 3, "stride": 3, "out_channels": 64},
                "final": {"in_channels": 64, "kernel_size": 1, "out_channels": 1}},
}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}     # MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip-level parameters);
                                                                         # bf16x3: three bf16 MFMAs per product
PEAK_HBM_GBS = 8000.0
FUSION_FAMILIES = ("conv3x3_bf16_128x128", "conv3x3_bf16_128x128+res", "conv3x3_bf16_128x64+res", "conv3x3_bf16_128x64")
TRAFFIC_JSON = "r03_traffic.json"
KERNEL_SOURCES = ("conv3x3_v6.hip", "conv3x3_v6_impl.h", "conv3x3_r64.hip", "stem.hip", "decoder.hip")


def synth_inputs(batch, views, size, device, seed):
    """PROBA-V-like synthetic LR stacks: a smooth-ish scene per sample + per-view jitter, uint16-quantised, in [0, 0.26]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    base = torch.rand((batch, 1, size // 8 + 1, size // 8 + 1), generator=g)
    base = torch.nn.functional.interpolate(base, size=(size, size), mode="bilinear", align_corners=True) * 0.25
    lrs = base + 0.01 * torch.rand((batch, views, size, size), generator=g)
    lrs = torch.round(lrs * 65535.0) / 65535.0
    return lrs.float().to(device), torch.ones((batch, views), dtype=torch.float32, device=device)


def flops_per_frame(views, size):
    px = size * size
    enc = 2.0 * px * (18 * 64 + 5 * 64 * 64 * 9)
    pair = 2.0 * px * (2 * 128 * 128 * 9 + 128 * 64 * 9)
    dec = 2.0 * px * (64 * 576 + 576)
    n, pairs = views, 0
    while n // 2 > 0:
        pairs += n // 2
        n //= 2
    return views * enc + pairs * pair + dec


def fusion_units(views):
    """algorithmic bytes of the recursive fusion = sum over levels of (2h read + h write) units of 64*HW*es."""
    n, units = views, 0
    while n // 2 > 0:
        units += 3 * (n // 2)
        n //= 2
    return units


def kernel_source_hash():
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "highres-net_amd", "hrnet_hip", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def precomputed_traffic(batch, views, size, precision):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes of tools/prof_all.sh (FETCH_SIZE doubled per the gfx950 note,
    + WRITE_SIZE), committed as profiles/r03_traffic.json together with a hash of the kernel sources they were taken on: stale or
    foreign numbers are not reported."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_JSON)))
        w = tj["workload"]
        if (w["batch"], w["views"], w["size"], w["precision"]) != (batch, views, size, precision):
            return None
        if tj.get("kernel_source_hash") != kernel_source_hash():
            return None
        return tj
    except (OSError, KeyError, ValueError):
        return None


def cpsnr_np(a, b):
    """Evaluator.cPSNR (Evaluator.py:34-38) with an all-ones mask: brightness-corrected PSNR of a against b, mean over the batch."""
    import numpy as np
    a = a.reshape(a.shape[0], -1).astype(np.float64)
    b = b.reshape(b.shape[0], -1).astype(np.float64)
    bias = (b - a).mean(1, keepdims=True)
    cmse = ((a + bias - b) ** 2).mean(1)
    return float((-10.0 * np.log10(np.maximum(cmse, 1e-300))).mean())


def cpu_baseline(nets, views, size, budget_s=20.0):
    """Time the torch-CPU port (oracle/torch_port.py) on a bounded sample of the same workload, and check both HIP paths against
    its output on that very sample (the oracle as the checker, never as the thing measured on the GPU side)."""
    from oracle import torch_port
    st = {k: v.detach().float().cpu() for k, v in nets["fp32"].state_dict().items()}
    threads = torch.get_num_threads()
    lrs, alphas = synth_inputs(1, views, size, "cpu", 5)
    t0 = time.perf_counter()
    torch_port.hrnet_forward(lrs, alphas, st)
    one = time.perf_counter() - t0                      # includes first-call warm-up
    n = int(max(1, min(8, budget_s // max(one, 1e-3))))
    lrs, alphas = synth_inputs(n, views, size, "cpu", 6)
    t0 = time.perf_counter()
    want = torch_port.hrnet_forward(lrs, alphas, st).numpy()
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 4), "unit": "SR frames/s", "cores": threads, "kind": "port",
           "sample": f"B={n} of the B=32 batch, n_views={views}, {size}x{size}->{3 * size}x{3 * size}, fp32, own torch-CPU port "
                     f"of HRNet.forward (oracle/torch_port.py), {threads} threads, {dt:.1f} s"}
    parity = {}
    with torch.no_grad():
        for name, net in nets.items():
            got = net(lrs.cuda(), alphas.cuda()).cpu().numpy()
            import numpy as np
            parity[name] = {"max_rel": float(np.abs(got - want).max() / np.abs(want).max()), "cpsnr_db": round(cpsnr_np(got, want), 2)}
    return out, parity


def timed(step, steps, warmup, device, hdist):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(device)
    hdist.barrier(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    hdist.barrier(device)
    return hdist.max_over_ranks(time.perf_counter() - t0, device)


def timed_local(step, steps, warmup, device):
    """Rank-local timing (no barrier): for the extra measurements only rank 0 takes."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0


def make_train_step(device, batch, views, patch, overlap=True, precision="fp32"):
    """The statements of src/train.py:164-191 on the HIP modules (random-init weights, synthetic batch)."""
    from DeepNetworks.HRNet import HRNet
    from DeepNetworks.ShiftNet import ShiftNet
    from hrnet_hip import losses
    from hrnet_hip.optim import FusedAdam
    torch.manual_seed(1234)
    fusion = HRNet(dict(NETWORK, precision=precision)).to(device).train()
    regis = ShiftNet().to(device).train()
    with torch.no_grad():
        regis.fc2.weight.normal_(0.0, 1e-3)                       # not the all-zero start: the registration branch does real work
    params = list(fusion.parameters()) + list(regis.parameters())
    opt = FusedAdam(params, lr=1e-4, overlap_early=list(regis.parameters()) if overlap else None)
    lrs, alphas = synth_inputs(batch, views, patch, device, seed=200 + int(os.environ.get("RANK", 0)))
    g = torch.Generator(device="cpu").manual_seed(7)
    hrs = (torch.rand((batch, 3 * patch, 3 * patch), generator=g) * 0.25).to(device)
    maps = torch.ones((batch, 3 * patch, 3 * patch), device=device)
    off = (3 * patch - 128) // 2

    def step(exchange=True):
        for f in opt._flat:
            f["buckets"].enabled = exchange                                                         # also silences the backward hook
        opt.zero_grad()
        srs = fusion(lrs, alphas)                                                                   # train.py:174
        ref = hrs[:, off:off + 128, off:off + 128].reshape(-1, 1, 128, 128)
        shifts = torch.stack([regis(torch.cat([ref, srs[:, :, off:off + 128, off:off + 128]], 1))], 1)   # register_batch :177-179
        b, n, h, w = srs.shape
        shifted = regis.transform(shifts.view(-1, 2), srs.view(-1, 1, h, w), device=device).view(-1, n, h, w)[:, 0]   # apply_shifts
        loss = -losses.get_loss(shifted, hrs, maps, metric="cPSNR", crop=3)                         # :183-185, crop mask folded in
        loss = torch.mean(loss) + 1e-6 * torch.mean(shifts) ** 2                                    # :186-187
        loss.backward()                                                                             # :190
        if exchange:
            step.early = opt._flat[0]["buckets"].early_launched_in_backward
            opt.allreduce()                                                                         # the one exchange step
        opt.step()                                                                                  # :191
        return loss
    step.early = False
    step.flat_parameters = opt._flat[0]["p"]                                                        # every weight of both models, one buffer
    step.initial = opt._flat[0]["p"].clone()
    return step


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N` launched directly (no torchrun environment): start N fresh child processes - one rank per GPU,
    rendezvous on 127.0.0.1 - BEFORE this process makes any GPU call (it never does: `torch.cuda.device_count()` does not
    initialise the device on this image), pass rank 0's stdout through and fail if any child fails.  Never an exec of a process
    that has touched the GPU."""
    import subprocess
    rehearsal = os.environ.get("HRN_DIST_BACKEND") == "gloo" and torch.cuda.device_count() == 0
    if not rehearsal and "HRN_BENCH_DEVICE" not in os.environ and torch.cuda.device_count() < n:
        raise SystemExit(f"bench.py --gpus {n}: only {torch.cuda.device_count()} ROCm device(s) visible - refusing to run fewer ranks than "
                         f"asked for (HRN_BENCH_DEVICE=<id> rehearses N ranks on one card; HRN_DIST_BACKEND=gloo on a box without a GPU "
                         f"rehearses the launcher alone)")
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for rank in range(n):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if rank == 0 else subprocess.DEVNULL))
        # wait for all of them; a rank that dies takes the others with it (they would sit in a collective until its timeout)
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes):
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                codes = [p.wait() for p in procs]
                break
            time.sleep(0.1)
        out0.seek(0)
        sys.stdout.write(out0.read().decode("utf-8", "replace"))
        sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py --gpus {n}: rank exit codes {codes}")


def ranks_seen(rank, ws, device):
    """Every rank's id as rank 0 sees them after an all_gather (proof that N processes took part), and the backend's name."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return [0], None
    t = torch.tensor([rank], dtype=torch.int64, device=device)
    got = [torch.zeros_like(t) for _ in range(ws)]
    dist.all_gather(got, t)
    return sorted(int(g.item()) for g in got), dist.get_backend()


def rehearse_plumbing(args, rank, ws, hdist):
    """No ROCm device + HRN_DIST_BACKEND=gloo: the launcher / rendezvous / barrier / max-over-ranks plumbing of the N-rank run on CPU
    stand-in steps (tests/test_dist_cpu.py::test_bench_spawns_its_own_ranks).  No kernel runs, `value` is null: never a measurement."""
    cpu = torch.device("cpu")
    seen, backend = ranks_seen(rank, ws, cpu)
    elapsed = timed_cpu(lambda: time.sleep(0.001), args.steps, args.warmup, hdist)
    if rank == 0:
        print(json.dumps({"metric": "SR frames/sec (384x384 out) at B=32, n_views=32", "value": None, "unit": "SR frames/s", "n_gpus": ws,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": None,
                          "data": "none: launcher rehearsal on CPU stand-in steps, no kernel ran", "rehearsal": True,
                          "ranks_seen": seen, "backend": backend, "mode": args.mode,
                          "config": {"workload": "plumbing only", "global_batch": args.batch * ws, "parallelism": f"dp{ws}"}}), flush=True)
    hdist.finalize()


def timed_cpu(step, steps, warmup, hdist):
    for _ in range(warmup):
        step()
    hdist.barrier(None)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    hdist.barrier(None)
    return hdist.max_over_ranks(time.perf_counter() - t0, None)


def extra(extras, key, fn):
    """One extra measurement: whatever goes wrong in it (an OOM on a box with less free HBM, ...) is recorded under its key and never
    costs the contract line."""
    try:
        extras[key] = fn()
    except Exception as e:                                     # noqa: BLE001 - the headline must survive any extra
        extras[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            torch.cuda.synchronize()
        except Exception:                                      # noqa: BLE001
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="forward", choices=["forward", "train"])
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--size", type=int, default=128, help="LR side (forward mode)")
    ap.add_argument("--patch", type=int, default=64, help="LR patch side (train mode; config.json patch_size)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--profile-steps", type=int, default=3, help="extra instrumented steps for the rooflines (not in `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / bf16x3 / config-2 / config-5 / train-step extra measurements")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                     # the driver's form of the command: this process only launches and collects
        return

    from hrnet_hip import dist as hdist
    rank, local_rank, ws = hdist.init()            # joins the process group BEFORE any GPU call of this process
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {ws}")
    if not torch.cuda.is_available():
        if os.environ.get("HRN_DIST_BACKEND") == "gloo":
            return rehearse_plumbing(args, rank, ws, hdist)
        raise SystemExit("bench.py needs a ROCm device: the HIP path has no CPU fallback")
    from hrnet_hip import binding
    from DeepNetworks.HRNet import HRNet
    device = torch.device("cuda", int(os.environ.get("HRN_BENCH_DEVICE", local_rank if ws > 1 else 0)))   # override: rehearsal of N ranks on one GPU
    torch.cuda.set_device(device)
    seen, backend = ranks_seen(rank, ws, device)
    if len(seen) != ws:
        raise SystemExit(f"all_gather saw ranks {seen}, expected {ws}")

    if args.mode == "train":
        tprec = "fp32" if args.precision == "bf16" else args.precision       # (bf16 storage has no training path: fp32 unless bf16x3 is asked for)
        step = make_train_step(device, args.batch, args.views, args.patch, precision=tprec)
        steps = max(1, min(args.steps, 10))
        warm = max(1, min(args.warmup, 3))
        elapsed = timed(step, steps, warm, device, hdist)
        no_x = timed(lambda: step(exchange=False), steps, 1, device, hdist) if ws > 1 else elapsed
        if rank == 0:
            print(json.dumps({
                "metric": f"train samples/sec of the src/train.py optimisation step (B={args.batch}/GPU, n_views={args.views}, {args.patch}x{args.patch} patches)",
                "value": round(args.batch * ws * steps / elapsed, 2), "unit": "samples/s", "n_gpus": ws, "steps": steps, "warmup": warm,
                "ms_per_step": round(elapsed / steps * 1e3, 2), "ms_per_step_without_exchange": round(no_x / steps * 1e3, 2),
                "early_slice_launched_during_backward": bool(step.early) if ws > 1 else None,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": tprec, "data": "synthetic",
                "ranks_seen": seen, "backend": backend,
                "config": {"workload": f"train.py:164-191 on the HIP modules: HRNet + ShiftNet + Lanczos + registered cPSNR loss + FusedAdam, "
                                       f"B={args.batch}/GPU, n_views={args.views}, {args.patch}x{args.patch} patches (BASELINE configs[3])",
                           "global_batch": args.batch * ws,
                           "parallelism": f"dp{ws}: {backend or 'no'} all-reduce of 139 MB of fp32 gradients per step, ShiftNet's slice overlapped with HRNet's backward",
                           "weights": "random init (seed 1234), fc2 ~ N(0, 1e-3)"}}), flush=True)
        hdist.finalize()
        return

    torch.manual_seed(1234)                               # same random-init weights on every rank
    net = HRNet(dict(NETWORK, precision=args.precision)).to(device).eval()
    lrs, alphas = synth_inputs(args.batch, args.views, args.size, device, seed=100 + rank)

    sr_holder = [None]

    def step():
        sr_holder[0] = net(lrs, alphas)                    # the module call the reference's callers make (predict.py:39, train.py:205)

    with torch.no_grad():
        elapsed = timed(step, args.steps, args.warmup, device, hdist)
    frames = args.batch * args.steps * ws
    value = frames / elapsed
    if rank != 0:
        hdist.finalize()
        return

    storage = {"bf16": "bf16 storage + fp32 accumulate", "fp32": "fp32 storage, exact-fp32 MFMA",
               "bf16x3": "fp32 values as bf16 hi+lo pairs, 3 bf16 MFMAs per product, fp32 accumulate"}[args.precision]
    tf = flops_per_frame(args.views, args.size) * value / 1e12
    line = {
        "metric": "SR frames/sec (384x384 out) at B=32, n_views=32", "value": round(value, 2), "unit": "SR frames/s",
        "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic", "ranks_seen": seen, "backend": backend,
        "config": {"workload": f"HRNet.forward B={args.batch}/GPU, n_views={args.views}, {args.size}x{args.size}->"
                               f"{3 * args.size}x{3 * args.size}, {storage} (BASELINE configs[2])",
                   "timed_call": "DeepNetworks.HRNet.HRNet.__call__ in eval mode -> torch.ops.hrnet_hip.hrnet_forward -> hrn_hrnet_forward (C ABI)",
                   "global_batch": args.batch * ws, "parallelism": f"dp{ws} replicas, no data-path collective",
                   "weights": "random init (torch default, seed 1234)"},
        "whole_forward_tflops": round(tf, 1),
        "roofline": None, "roofline_hbm": None, "parity": {}, "cpu_baseline": None, "kernels": {},
    }
    try:
        with torch.no_grad():
            measure_rank0(args, ws, device, net, lrs, alphas, step, sr_holder, binding, line)
    except Exception as e:                                     # noqa: BLE001 - the contract line is printed whatever the diagnostics do
        line["diagnostics_error"] = f"{type(e).__name__}: {e}"[:400]
    finally:
        print(json.dumps(line), flush=True)
    hdist.finalize()


def profile_families(binding, device, fn, n=1):
    """Run fn() n times with the library's hipEvent profiler on; -> {family: {launches, ms, flops, bytes}} summed over the n runs."""
    binding.profile_enable(True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize(device)
    binding.profile_enable(False)
    return binding.profile_read()


def measure_rank0(args, ws, device, net, lrs, alphas, step, sr_holder, binding, line):
    """Everything of the line beyond the contract keys (rank 0, after the timed region): rooflines, parity, extras, CPU baseline."""
    from DeepNetworks.HRNet import HRNet
    nprof = max(1, args.profile_steps)
    prof = profile_families(binding, device, step, nprof)
    sr = sr_holder[0]
    kernels = line["kernels"]
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        avg = v["ms"] / max(v["launches"], 1)
        kernels[name] = {"launches_per_step": v["launches"] // nprof, "avg_ms": round(avg, 4),
                         "share": round(v["ms"] / total_ms, 4),
                         "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["ms"] > 0 else None,
                         "gbs": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else None}
    name, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
    achieved = v["flops"] / v["ms"] / 1e9            # TFLOP/s: algorithmic FLOPs of the launches / their summed duration
    peak = PEAK_TFLOPS[args.precision]
    tj = precomputed_traffic(args.batch, args.views, args.size, args.precision)
    traffic = tj["bytes_per_launch"].get(name) if tj else None
    line["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic,
                        "traffic_source": f"precomputed: profiles/{TRAFFIC_JSON} (rocprofv3 --pmc passes of tools/prof_all.sh on these kernel sources)" if traffic else None,
                        "bytes_per_launch_algorithmic": v["bytes"] / max(v["launches"], 1),
                        "launches": v["launches"], "avg_launch_ms": round(v["ms"] / max(v["launches"], 1), 4),
                        "flops_per_launch": v["flops"] / max(v["launches"], 1),
                        "algorithmic_gbs": round(v["bytes"] / v["ms"] / 1e6, 1)}
    if args.precision == "bf16x3":
        line["roofline"]["note"] = "peak = the bf16 MFMA peak / 3: every product costs three bf16 MFMAs (hi*hi + hi*lo + lo*hi)"
    # the recursive fusion as ONE operator against the HBM roof (north_star): SURVEY 8d's algorithmic bytes = units of
    # 64*HW*es (2h read + h write per level) per sample; measured time = all its conv launches of the instrumented steps
    fus = [prof[f] for f in FUSION_FAMILIES if f in prof]
    if fus and args.precision == "bf16":
        fus_ms = sum(f["ms"] for f in fus) / nprof
        alg = fusion_units(args.views) * 64 * args.size * args.size * 2 * args.batch
        layer_bytes = sum(f["bytes"] for f in fus) / nprof
        ctr = sum(tj["bytes_per_launch"].get(k, 0) * kernels[k]["launches_per_step"] for k in FUSION_FAMILIES if tj and k in kernels) if tj else None
        line["roofline_hbm"] = {"stage": "recursive fusion (all levels of one forward)", "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                                "achieved": round(alg / fus_ms / 1e6, 1), "frac": round(alg / fus_ms / 1e6 / PEAK_HBM_GBS, 4),
                                "algorithmic_bytes": alg, "ms": round(fus_ms, 3),
                                "per_layer_algorithmic_bytes": layer_bytes, "per_layer_algorithmic_gbs": round(layer_bytes / fus_ms / 1e6, 1),
                                "traffic": ctr, "traffic_gbs": round(ctr / fus_ms / 1e6, 1) if ctr else None,
                                "note": "the stage is MFMA-bound (~1000 FLOP/B): its HBM fraction states how far the conv-per-launch "
                                        "decomposition is from the fused operator's minimal traffic, not a saturated memory system"}
    enc = [prof[f] for f in prof if f.startswith("conv3x3_") and "_64x64" in f]
    if enc:
        ems, efl = sum(f["ms"] for f in enc), sum(f["flops"] for f in enc)
        eby = sum(f["bytes"] for f in enc)
        ectr = sum(tj["bytes_per_launch"].get(k, 0) * kernels[k]["launches_per_step"] for k in kernels if "_64x64" in k) if tj else None
        line["roofline_encoder"] = {"stage": "encoder 64->64 convs (north_star: >= 0.70 of the MFMA roof)", "bound": "mfma", "unit": "TFLOP/s",
                                    "achieved": round(efl / ems / 1e9, 1), "peak": peak, "frac": round(efl / ems / 1e9 / peak, 4),
                                    "ms": round(ems / nprof, 3),
                                    # 288 FLOP/B: the stage sits at the ridge, and on this part it is the memory side it runs into first
                                    "hbm": {"achieved": round(eby / ems / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                            "frac": round(eby / ems / 1e6 / PEAK_HBM_GBS, 4), "algorithmic_bytes": eby / nprof,
                                            "traffic": ectr, "traffic_gbs": round(ectr / (ems / nprof) / 1e6, 1) if ectr else None}}

    # ---- parity of the run that was timed: against the exact-fp32 HIP path on the same batch (cPSNR: Evaluator.py:34-38, all-ones mask)
    parity, extras = line["parity"], {}
    nets = {args.precision: net}

    def other_net(prec):
        if prec not in nets:
            n2 = HRNet(dict(NETWORK, precision=prec)).to(device).eval()
            n2.load_state_dict(net.state_dict())
            nets[prec] = n2
        return nets[prec]

    ref32 = sr if args.precision == "fp32" else other_net("fp32")(lrs, alphas)
    ones = torch.ones_like(sr[:, 0])

    def vs_fp32(x):
        return {"max_rel": float((x - ref32).abs().max() / ref32.abs().max()),
                "cpsnr_db": round(float(binding.get_loss(x[:, 0], ref32[:, 0], ones, "cPSNR").mean()), 2),
                "batch": f"the timed batch (B={args.batch}, n_views={args.views})"}
    if args.precision != "fp32":
        parity[f"{args.precision}_vs_fp32_hip_path"] = vs_fp32(sr)
    else:
        parity["bf16_vs_fp32_hip_path"] = vs_fp32(other_net("bf16")(lrs, alphas))

    if not args.no_extras and ws == 1:             # (multi-rank runs keep rank 0's tail short: the other ranks are already done)
        def path_extra(prec, fam, steps):
            n2 = other_net(prec)
            out = [None]

            def f():
                out[0] = n2(lrs, alphas)
            t = timed_local(f, steps, 1, device) / steps
            pf = profile_families(binding, device, f, 1)
            r = {"frames_per_s": round(args.batch / t, 1), "ms_per_step": round(t * 1e3, 2), "steps": steps,
                 "workload": f"B={args.batch}, n_views={args.views}, precision={prec}"}
            if prec != "fp32":
                r["parity_vs_fp32_hip_path"] = vs_fp32(out[0])
            if fam in pf and pf[fam]["ms"] > 0:
                a = pf[fam]["flops"] / pf[fam]["ms"] / 1e9
                r["roofline"] = {"kernel": fam, "bound": "mfma", "achieved": round(a, 2), "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                                 "frac": round(a / PEAK_TFLOPS[prec], 4), "avg_launch_ms": round(pf[fam]["ms"] / pf[fam]["launches"], 4)}
            tot = sum(v["flops"] for v in pf.values()) / max(sum(v["ms"] for v in pf.values()), 1e-9) / 1e9
            r["whole_forward_tflops"] = round(tot, 1)
            return r
        if args.precision != "fp32":        # the exact-fp32 path at the metric's shape (the path that meets the 1e-3 contract bit for bit)
            extra(extras, "fp32_path", lambda: path_extra("fp32", "conv3x3_f32_128x128", 3))
        if args.precision != "bf16x3" and binding.has_bf16x3():
            extra(extras, "bf16x3_path", lambda: path_extra("bf16x3", "conv3x3_bf16x3_128x128+res", 5))

        def c2():
            l2, a2 = synth_inputs(16, 16, 128, device, seed=300)
            n32 = other_net("fp32")
            t = timed_local(lambda: n32(l2, a2), 5, 2, device) / 5
            return {"frames_per_s": round(16 / t, 1), "ms_per_step": round(t * 1e3, 2), "steps": 5,
                    "workload": "BASELINE configs[1]: B=16, n_views=16, 128x128->384x384, fp32"}
        extra(extras, "c2_fp32", c2)

        def lanczos_roof():
            def at(side):
                img = torch.rand((1, args.batch, side, side), device=device)
                sh = (torch.rand((args.batch, 2), device=device) - 0.5) * 2
                f = lambda: binding.lanczos_shift(img, sh)         # noqa: E731
                timed_local(f, 3, 2, device)
                pf = profile_families(binding, device, f, 10)["lanczos_shift"]
                gbs = pf["bytes"] / pf["ms"] / 1e6
                return {"achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4), "avg_launch_ms": round(pf["ms"] / pf["launches"], 4),
                        "algorithmic_bytes_per_launch": pf["bytes"] / pf["launches"],
                        "workload": f"lanczos_shift of {args.batch} images of {side}x{side} (one launch; SURVEY 8d: 2*B*9HW*4 B)"}
            r = {"kernel": "lanczos_shift", "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s"}
            r.update(at(3 * args.size))
            r["at_config5_size"] = at(3 * 512)                     # BASELINE configs[4]: 1536 x 1536 (the metric's 384 x 384 is 38 MB: a launch)
            return r
        extra(extras, "roofline_lanczos", lanczos_roof)

        def host_buffers():
            """The boundary with HOST buffers (never `value`): lrs / alphas in pinned host memory in, the SR frames in pinned host memory
            out - serially on one stream, and double-buffered with the copies on their own HIP streams beside the kernels."""
            n_it = 10
            h_lrs, h_al = lrs.cpu().pin_memory(), alphas.cpu().pin_memory()
            h_sr = torch.empty((args.batch, 1, 3 * args.size, 3 * args.size)).pin_memory()

            def serial():
                d_l, d_a = h_lrs.to(device, non_blocking=True), h_al.to(device, non_blocking=True)
                h_sr.copy_(net(d_l, d_a), non_blocking=True)
            t_serial = timed_local(serial, n_it, 2, device) / n_it
            s_in, s_out, s_main = torch.cuda.Stream(device), torch.cuda.Stream(device), torch.cuda.current_stream(device)
            d_in = [(torch.empty_like(lrs), torch.empty_like(alphas)) for _ in range(2)]
            d_out = [torch.empty_like(h_sr, device=device) for _ in range(2)]
            h_out = [torch.empty_like(h_sr).pin_memory() for _ in range(2)]
            ev_in = [torch.cuda.Event() for _ in range(2)]
            ev_done = [torch.cuda.Event() for _ in range(2)]
            ev_free = [torch.cuda.Event() for _ in range(2)]

            def upload(i):
                b = i & 1
                with torch.cuda.stream(s_in):
                    s_in.wait_event(ev_free[b])                     # the forward that read this input pair has run
                    d_in[b][0].copy_(h_lrs, non_blocking=True)
                    d_in[b][1].copy_(h_al, non_blocking=True)
                    ev_in[b].record(s_in)

            def pipelined(n):
                for b in range(2):
                    ev_free[b].record(s_main)
                upload(0)
                for i in range(n):
                    b = i & 1
                    if i + 1 < n:
                        upload(i + 1)
                    s_main.wait_event(ev_in[b])
                    s_main.wait_event(ev_done[b])                   # the download that read d_out[b] two steps ago has run
                    d_out[b].copy_(net(d_in[b][0], d_in[b][1]))
                    ev_free[b].record(s_main)
                    with torch.cuda.stream(s_out):
                        s_out.wait_stream(s_main)
                        h_out[b].copy_(d_out[b], non_blocking=True)
                        ev_done[b].record(s_out)
                torch.cuda.synchronize(device)
            for b in range(2):
                ev_done[b].record(s_out)
            pipelined(3)
            t0 = time.perf_counter()
            pipelined(n_it)
            t_pipe = (time.perf_counter() - t0) / n_it
            return {"serial_frames_per_s": round(args.batch / t_serial, 1), "serial_ms_per_step": round(t_serial * 1e3, 3),
                    "overlapped_frames_per_s": round(args.batch / t_pipe, 1), "overlapped_ms_per_step": round(t_pipe * 1e3, 3),
                    "bytes_in": lrs.numel() * 4 + alphas.numel() * 4, "bytes_out": h_sr.numel() * 4,
                    "note": "pinned host buffers; overlapped = H2D, kernels and D2H on three HIP streams, inputs and outputs double-buffered"}
        extra(extras, "host_buffers", host_buffers)

        def shiftnet_fwd():
            from DeepNetworks.ShiftNet import ShiftNet
            sn = ShiftNet().to(device).eval()
            x = torch.rand((args.batch, 2, 128, 128), device=device)
            f = lambda: sn(x)                                   # noqa: E731
            t = timed_local(f, 5, 2, device) / 5
            pf = profile_families(binding, device, f, 3)
            r = {"ms": round(t * 1e3, 3), "workload": f"ShiftNet.forward (eval), B={args.batch} pairs of 128x128",
                 "kernels": {k: {"avg_ms": round(v["ms"] / v["launches"], 4), "launches_per_forward": v["launches"] // 3,
                                 "tflops": round(v["flops"] / v["ms"] / 1e9, 2), "gbs": round(v["bytes"] / v["ms"] / 1e6, 1)}
                             for k, v in sorted(pf.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}}
            if "fc1" in pf:
                g = pf["fc1"]["bytes"] / pf["fc1"]["ms"] / 1e6
                r["fc1_roofline"] = {"bound": "hbm", "achieved": round(g, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(g / PEAK_HBM_GBS, 4),
                                     "note": "one read of the 134 MB weight matrix per forward"}
            return r
        extra(extras, "shiftnet_forward", shiftnet_fwd)
        step()                                   # leave the timed path's output the most recent one again

        if args.precision == "bf16":
            def c5():
                # BASELINE configs[4]: the same forward on 512 x 512 tiles (16 x the pixels: 3 x 32 GiB of workspace)
                binding._ws_cache.clear()
                torch.cuda.empty_cache()
                free, _ = torch.cuda.mem_get_info(device)
                if free < 120 * (1 << 30):
                    return {"skipped": f"only {free / (1 << 30):.0f} GiB of HBM free; the 512x512 forward needs ~100 GiB of workspace"}
                l5, a5 = synth_inputs(32, 32, 512, device, seed=500)
                try:
                    t = timed_local(lambda: net(l5, a5), 2, 1, device) / 2
                finally:
                    del l5, a5
                    binding._ws_cache.clear()
                    torch.cuda.empty_cache()
                return {"frames_per_s": round(32 / t, 1), "ms_per_step": round(t * 1e3, 1), "steps": 2,
                        "workload": "BASELINE configs[4]: B=32, n_views=32, 512x512->1536x1536, bf16"}
            extra(extras, "c5_bf16", c5)

        def train(prec):
            binding._ws_cache.clear()
            torch.cuda.empty_cache()
            tstep = make_train_step(device, 32, 32, 64, precision=prec)
            with torch.enable_grad():
                t = timed_local(tstep, 3, 2, device) / 3
                pf = profile_families(binding, device, tstep, 1)
            top = sorted(pf.items(), key=lambda kv: -kv[1]["ms"])[:6]
            del tstep
            return {"ms_per_step": round(t * 1e3, 1), "samples_per_s": round(32 / t, 1), "steps": 3,
                    "top_kernel_families_ms": {k: round(v["ms"], 2) for k, v in top},
                    "workload": f"src/train.py:164-191 on the HIP modules, B=32, n_views=32, 64x64 patches, HRNet in {prec} "
                                f"(python bench.py --mode train{'' if prec == 'fp32' else ' --precision ' + prec})"}
        extra(extras, "train_step", lambda: train("fp32"))
        if binding.has_bf16x3():
            extra(extras, "train_step_bf16x3", lambda: train("bf16x3"))
    line.update(extras)

    if ws == 1 and not args.no_cpu_baseline:
        binding._ws_cache.clear()
        torch.cuda.empty_cache()
        cpu, vs_port = cpu_baseline({p: other_net(p) for p in (["bf16", "fp32"] + (["bf16x3"] if binding.has_bf16x3() else []))},
                                    args.views, args.size)
        line["cpu_baseline"] = cpu
        parity["vs_cpu_port_on_cpu_baseline_sample"] = vs_port


if __name__ == "__main__":
    main()
