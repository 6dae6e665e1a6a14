"""Fused Adam for the joint HRNet + ShiftNet update.

Drop-in for the reference's `optim.Adam(list(fusion_model.parameters()) + list(regis_model.parameters()), lr=...)`
(src/train.py:252) including `lr_scheduler.ReduceLROnPlateau` on top of it (train.py:253): same constructor arguments,
`param_groups`, `zero_grad()`, `step()`, `state_dict()` surface, torch.optim.Adam's arithmetic (no amsgrad).

What is different underneath: every parameter of a group is re-homed into ONE flat fp32 device buffer (the parameters
become views of it), their gradients live in one flat buffer as well (`p.grad` are views), and `step()` is a single
`hrn_adam_step` launch per group instead of ~10 framework kernels per tensor.  The flat gradient buffer is also what
the data-parallel exchange reduces: `allreduce()` issues one all-reduce per group (139 MB for the two models - on xGMI a
few large messages beat many small ones), between `loss.backward()` and `step()`.
"""
import torch
import torch.distributed as dist

from . import binding


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
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
            self._flat.append(dict(params=ps, p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p), step=0))
        binding.bump_param_epoch()

    def zero_grad(self, set_to_none=False):
        """Zero the flat gradient buffers; `.grad` stays a view of them (set_to_none is ignored on purpose)."""
        for f in self._flat:
            if f is None:
                continue
            f["g"].zero_()
            self._rebind(f)

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
        """Average the flat gradient buffers over the process group (identity without one).  Returns bytes reduced."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return 0
        total = 0
        works = []
        for f in self._flat:
            if f is None:
                continue
            self._rebind(f)
            works.append((f, dist.all_reduce(f["g"], op=dist.ReduceOp.SUM, async_op=True)))
            total += f["g"].numel() * 4
        for f, w in works:
            w.wait()
            f["g"].div_(dist.get_world_size())
        return total

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            self._rebind(f)
            f["step"] += 1
            b1, b2 = group["betas"]
            binding.adam_step(f["p"], f["g"], f["m"], f["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"], f["step"])
        return loss
