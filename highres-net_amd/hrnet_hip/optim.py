"""Fused Adam for the joint HRNet + ShiftNet update.

Drop-in for the reference's `optim.Adam(list(fusion_model.parameters()) + list(regis_model.parameters()), lr=...)`
(src/train.py:252) including `lr_scheduler.ReduceLROnPlateau` on top of it (train.py:253): same constructor arguments,
`param_groups`, `zero_grad()`, `step()`, `state_dict()` surface, torch.optim.Adam's arithmetic (no amsgrad).

What is different underneath: every parameter of a group is re-homed into ONE flat fp32 device buffer (the parameters
become views of it), their gradients live in one flat buffer as well (`p.grad` are views), and `step()` is a single
`hrn_adam_step` launch per group instead of ~10 framework kernels per tensor.  The flat gradient buffer is also what
the data-parallel exchange reduces (`allreduce()`, between `loss.backward()` and `step()`): with `overlap_early=regis_model.parameters()`
ShiftNet's 137 MB slice goes on the wire from a backward hook as soon as `hrn_shiftnet_backward` has written it, under the whole
HRNet backward pass; `allreduce()` then reduces HRNet's 2.4 MB and waits (hrnet_hip.dist.GradBuckets).
"""
import torch

from . import binding


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, overlap_early=None):
        """overlap_early: parameters (consecutive in `params`, e.g. `regis_model.parameters()`) whose gradients autograd completes
        first: their slice of the flat gradient buffer is all-reduced from a backward hook while the rest of the backward pass
        still runs (hrnet_hip.dist.GradBuckets); `allreduce()` then only has the remainder left."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        early_ids = {id(p) for p in (overlap_early or [])}
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            for p in ps:
                if not p.is_cuda or p.device != dev or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam needs float32 parameters on one ROCm device (no CPU fallback)")
            n = sum(p.numel() for p in ps)
            n_pad = (n + 3) // 4 * 4
            flat_p = torch.zeros(n_pad, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(n_pad, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in ps:
                    k = p.numel()
                    flat_p[off:off + k].copy_(p.reshape(-1))
                    p.data = flat_p[off:off + k].view(p.shape)          # the parameter now lives in the flat buffer
                    p.grad = flat_g[off:off + k].view(p.shape)
                    off += k
            from .dist import GradBuckets
            buckets = GradBuckets(flat_g, ps, [p for p in ps if id(p) in early_ids])
            self._flat.append(dict(params=ps, p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p), step=0, buckets=buckets))
        binding.bump_param_epoch()

    def zero_grad(self, set_to_none=False):
        """Zero the flat gradient buffers; `.grad` stays a view of them (set_to_none is ignored on purpose)."""
        for f in self._flat:
            if f is None:
                continue
            f["g"].zero_()
            self._rebind(f)
            f["buckets"].begin()

    @staticmethod
    def _rebind(f):
        off = 0
        for p in f["params"]:
            k = p.numel()
            view = f["g"][off:off + k].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                if p.grad is not None:
                    view.add_(p.grad)                                   # someone replaced .grad: fold it back in
                p.grad = view
            off += k

    def allreduce(self):
        """Average the flat gradient buffers over the process group (identity without one): waits for the slice a backward hook has
        already put on the wire (`overlap_early`) and reduces the rest.  Call between backward() and step() - step() does it itself
        if the caller did not (the stock train.py loop with only the constructor swapped).  Returns bytes reduced."""
        total = 0
        for f in self._flat:
            if f is None:
                continue
            b = f["buckets"]
            if b.early_launched_in_backward:
                b.wait_early()                # nothing may be folded into a slice that is still on the wire
            self._rebind(f)
            total += b.finish()
        return total

    def close(self):
        """Detach from the parameters' backward hooks (call before building another optimiser over the same parameters)."""
        for f in self._flat:
            if f is not None:
                f["buckets"].close()

    def __del__(self):
        try:
            self.close()
        except Exception:                     # noqa: BLE001 - interpreter shutdown
            pass

    # -- torch.optim.Adam's checkpoint surface: per-parameter `step`, `exp_avg`, `exp_avg_sq` (the moments live in flat buffers here)
    def state_dict(self):
        self.state.clear()
        for f in self._flat:
            if f is None:
                continue
            off = 0
            for p in f["params"]:
                k = p.numel()
                self.state[p] = {"step": torch.tensor(float(f["step"])), "exp_avg": f["m"][off:off + k].view(p.shape).clone(),
                                 "exp_avg_sq": f["v"][off:off + k].view(p.shape).clone()}
                off += k
        sd = super().state_dict()
        self.state.clear()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)                  # param_groups (lr, betas, ...) + per-parameter state, cast to the parameters
        for f in self._flat:
            if f is None:
                continue
            off, steps = 0, set()
            for p in f["params"]:
                k = p.numel()
                st = self.state.get(p)
                if st:
                    f["m"][off:off + k].copy_(st["exp_avg"].reshape(-1))
                    f["v"][off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                off += k
            if len(steps) > 1:
                raise ValueError("FusedAdam keeps one step count per parameter group; the checkpoint holds several")
            if steps:
                f["step"] = steps.pop()
        self.state.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            b = f["buckets"]
            if b._active() and not b._reduced:      # the caller did not run the exchange: do it here rather than step on local gradients
                if b.early_launched_in_backward:
                    b.wait_early()
                self._rebind(f)
                b.finish()
            self._rebind(f)
            f["step"] += 1
            b1, b2 = group["betas"]
            torch.ops.hrnet_hip.adam_step(f["p"], f["g"], f["m"], f["v"], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                          float(group["weight_decay"]), int(f["step"]))
        return loss
