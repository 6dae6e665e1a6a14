#!/usr/bin/env python3
"""bench.py - SR frames/s of the MI355X HighRes-net forward (BASELINE.json metric) + roofline + CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE HRNet.forward(lrs, alphas) on one resident batch of the workload the metric is quoted on
(BASELINE.json configs[2]: B=32, n_views=32, 128x128 -> 384x384, bf16 storage / fp32 accumulation).  With N GPUs
every rank runs its own batch (weak scaling, no data-path collective: SURVEY.md section 8e); `value` = N*B*K / max-over-ranks
wall time.  Rank 0 prints one JSON line with the contract keys plus `roofline` (dominant kernel, measured live with
the library's hipEvent profiler over extra instrumented steps) and `cpu_baseline` (own torch-CPU port of the same
forward on a bounded sample of the same workload, host cores stated; N=1 only).
"""
import argparse
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
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}     # MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


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


def cpu_baseline(views, size, budget_s=20.0):
    """Time the torch-CPU port (oracle/torch_port.py) on a bounded sample of the same workload."""
    from oracle import torch_port, weights
    st = weights.to_torch_state(weights.hrnet_state(1234))
    threads = torch.get_num_threads()
    lrs, alphas = synth_inputs(1, views, size, "cpu", 5)
    t0 = time.perf_counter()
    torch_port.hrnet_forward(lrs, alphas, st)
    one = time.perf_counter() - t0                      # includes first-call warm-up
    n = int(max(1, min(8, budget_s // max(one, 1e-3))))
    lrs, alphas = synth_inputs(n, views, size, "cpu", 6)
    t0 = time.perf_counter()
    torch_port.hrnet_forward(lrs, alphas, st)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "SR frames/s", "cores": threads, "kind": "port",
            "sample": f"B={n} of the B=32 batch, n_views={views}, {size}x{size}->{3 * size}x{3 * size}, fp32, own torch-CPU port "
                      f"of HRNet.forward (oracle/torch_port.py), {threads} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--profile-steps", type=int, default=3, help="extra instrumented steps for the roofline (not in `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from hrnet_hip import binding, dist as hdist
    from DeepNetworks.HRNet import HRNet

    rank, local_rank, ws = hdist.init()
    if ws != args.gpus and ws > 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {ws}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the HIP path has no CPU fallback")
    device = torch.device("cuda", local_rank if ws > 1 else 0)
    torch.cuda.set_device(device)

    torch.manual_seed(1234)                               # same random-init weights on every rank
    net = HRNet(dict(NETWORK, precision=args.precision)).to(device).eval()
    lrs, alphas = synth_inputs(args.batch, args.views, args.size, device, seed=100 + rank)
    sr = torch.empty((args.batch, 1, 3 * args.size, 3 * args.size), dtype=torch.float32, device=device)
    packed, dt = net.packed_parameters()

    def step():
        binding.hrnet_forward(packed, dt, 2, True, lrs, alphas, out=sr)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(device)
        hdist.barrier(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(device)
        hdist.barrier(device)
        elapsed = time.perf_counter() - t0
    elapsed = hdist.max_over_ranks(elapsed, device)
    frames = args.batch * args.steps * ws
    value = frames / elapsed

    roofline, kernels = None, {}
    if rank == 0:
        # instrumented steps: hipEvent pair around every kernel launch, on the launch stream (torch's current stream)
        binding.profile_enable(True)
        with torch.no_grad():
            for _ in range(max(1, args.profile_steps)):
                step()
        torch.cuda.synchronize(device)
        binding.profile_enable(False)
        prof = binding.profile_read()
        total_ms = sum(v["ms"] for v in prof.values()) or 1.0
        for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            avg = v["ms"] / max(v["launches"], 1)
            kernels[name] = {"launches_per_step": v["launches"] // max(1, args.profile_steps), "avg_ms": round(avg, 4),
                             "share": round(v["ms"] / total_ms, 4),
                             "tflops": round(v["flops"] / v["ms"] / 1e9, 2) if v["ms"] > 0 else None,
                             "gbs": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else None}
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        name, v = dom
        achieved = v["flops"] / v["ms"] / 1e9            # TFLOP/s: algorithmic FLOPs of the launches / their summed duration
        peak = PEAK_TFLOPS[args.precision]
        # HBM bytes per (average) launch come from separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950
        # note, + WRITE_SIZE) committed in profiles/r01_traffic.json; null when this run is not that workload
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            w = tj["workload"]
            if (w["batch"], w["views"], w["size"], w["precision"]) == (args.batch, args.views, args.size, args.precision):
                traffic = tj["bytes_per_launch"].get(name)
        except (OSError, KeyError, ValueError):
            traffic = None
        roofline = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "bytes_per_launch_algorithmic": v["bytes"] / max(v["launches"], 1),
                    "launches": v["launches"], "avg_launch_ms": round(v["ms"] / max(v["launches"], 1), 4),
                    "flops_per_launch": v["flops"] / max(v["launches"], 1),
                    "algorithmic_gbs": round(v["bytes"] / v["ms"] / 1e6, 1)}

    cpu = None
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.views, args.size)

    if rank == 0:
        tf = flops_per_frame(args.views, args.size) * value / 1e12
        line = {
            "metric": "SR frames/sec (384x384 out) at B=32, n_views=32", "value": round(value, 2), "unit": "SR frames/s",
            "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"HRNet.forward B={args.batch}/GPU, n_views={args.views}, {args.size}x{args.size}->"
                                   f"{3 * args.size}x{3 * args.size}, {args.precision} storage + fp32 accumulate (BASELINE configs[2])",
                       "global_batch": args.batch * ws, "parallelism": f"dp{ws} replicas, no data-path collective",
                       "weights": "random init (torch default, seed 1234)"},
            "whole_forward_tflops": round(tf, 1),
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(line), flush=True)
    hdist.finalize()


if __name__ == "__main__":
    main()
