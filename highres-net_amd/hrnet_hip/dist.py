"""One-process-per-GPU plumbing for the data-parallel forward.

HRNet.forward has no cross-sample operation, so the batch shards over ranks with NO data-path collective:
every rank runs the same kernels on its own B samples ("replicas"; weak scaling).  torch.distributed (backend
"nccl" == RCCL on ROCm, "gloo" in the CPU tests) is used only for the rendezvous, the barriers around the timed
region and the max-over-ranks reduction of the elapsed time.

Training (row f3) adds ONE exchange step per optimisation step: the average of the parameter gradients over ranks
(`allreduce_gradients`, called between `loss.backward()` and `optimizer.step()`, train.py:190-191).  HRNet has 0.59 M
parameters (2.4 MB) and ShiftNet 34.2 M (137 MB, almost all of it fc1): gradients are flattened into buckets of
`bucket_mb` and each bucket is all-reduced once - on xGMI a ring all-reduce is per-link bound (~153 GB/s), so few large
messages beat many small ones; the first buckets are on the wire while later ones are still being flattened.
"""
import os

import torch
import torch.distributed as dist


def world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched directly."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Join the process group when WORLD_SIZE > 1.  Returns (rank, local_rank, world_size)."""
    rank, local_rank, ws = world()
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # HRN_DIST_BACKEND=gloo: rehearsal of the multi-rank paths on a box with fewer GPUs than ranks (gloo moves CUDA tensors too)
            backend = os.environ.get("HRN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, local_rank, ws


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (identity without a process group)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shard(global_batch, rank, world_size):
    """Contiguous sample range [lo, hi) of rank `rank` (rank r takes samples [r*B, (r+1)*B))."""
    if global_batch % world_size != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world_size}")
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def allreduce_gradients(modules, bucket_mb=64):
    """Average `.grad` of every parameter of `modules` (a module or an iterable of modules) over all ranks, in place.

    Identity without a process group.  Parameters are walked in registration order, which is identical on every rank, so
    the buckets line up; parameters without a gradient on this rank contribute zeros (and receive the average).  Returns
    the number of bytes reduced (for logging).
    """
    if isinstance(modules, torch.nn.Module):
        modules = [modules]
    params = [p for m in modules for p in m.parameters() if p.requires_grad]
    if not dist.is_initialized() or dist.get_world_size() == 1 or not params:
        return 0
    ws = dist.get_world_size()
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    limit = int(bucket_mb * (1 << 20))
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.grad.numel() * p.grad.element_size()
        if cur and (cur_bytes + nbytes > limit or p.grad.dtype != cur[0].grad.dtype or p.grad.device != cur[0].grad.device):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    pending, total = [], 0
    for b in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in b])
        total += flat.numel() * flat.element_size()
        pending.append((b, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
    for b, flat, work in pending:
        work.wait()
        flat.div_(ws)
        off = 0
        for p in b:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return total


class GradBuckets:
    """Overlapped averaging of ONE flat gradient buffer over the ranks (the exchange step of a data-parallel train step,
    SURVEY.md sections 5 and 8e).

    `params` are the parameters whose `.grad` are consecutive views of `flat` (the layout hrnet_hip.optim.FusedAdam builds);
    `early` is a contiguous run of them whose gradients autograd finishes FIRST - for `src/train.py` that is ShiftNet
    (`loss -> lanczos -> ShiftNet -> HRNet`): its 137 MB, 98 % of the message, are complete while the whole backward pass of HRNet
    is still to run.  A post-accumulate hook on every early parameter counts them in; the last one puts the early slice on the
    wire (`all_reduce(..., async_op=True)`: RCCL works on its own stream behind the kernels already queued), and `finish()`,
    called between `loss.backward()` and `optimizer.step()`, reduces what is left (HRNet: 2.4 MB), waits, and divides by the
    world size.  Without a process group everything is the identity.  Works on any device (the CPU tests drive it on gloo)."""

    def __init__(self, flat, params, early=()):
        import weakref
        self.flat, self.params = flat, list(params)
        early = list(early)
        ids = {id(p): i for i, p in enumerate(self.params)}
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += p.numel()
        self.n = off
        self.lo = self.hi = 0
        self._early = early
        self._early_n = len(early)
        self._handles = []
        if early:
            idx = sorted(ids[id(p)] for p in early)
            if idx != list(range(idx[0], idx[0] + len(idx))):
                raise ValueError("GradBuckets: the early parameters must be consecutive in the flat buffer")
            self.lo, self.hi = offs[idx[0]], offs[idx[-1]] + self.params[idx[-1]].numel()
            ref = weakref.ref(self)            # the hook must not keep a replaced bucket (and its 139 MB buffer) alive

            def hook(param, ref=ref):
                me = ref()
                if me is not None:
                    me._arrived(param)
            for p in early:
                self._handles.append(p.register_post_accumulate_grad_hook(hook))
        self._count, self._work, self.early_launched_in_backward, self._reduced = 0, [], False, False
        self.enabled = True                 # False: a step without the exchange (bench.py's "without exchange" leg)

    def close(self):
        """Remove the backward hooks (a rebuilt optimiser registers its own; stale ones would reduce a dead buffer)."""
        for h in self._handles:
            h.remove()
        self._handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:                   # noqa: BLE001 - interpreter shutdown
            pass

    def begin(self):
        """Start of a step (optimizer.zero_grad()): forget the previous step's counts."""
        self._count, self._work, self.early_launched_in_backward, self._reduced = 0, [], False, False

    def _active(self):
        return self.enabled and dist.is_initialized() and dist.get_world_size() > 1

    def _early_grads_in_flat(self):
        """True when every early parameter's .grad still IS its slice of the flat buffer.  After `module.zero_grad()` (torch's
        set_to_none=True) or `p.grad = None` autograd installs fresh tensors: the slice then holds nothing of this step and must
        not go on the wire from the hook - finish() reduces it after the optimiser has folded the gradients back in."""
        lo = self.flat.data_ptr()
        hi = lo + self.flat.numel() * self.flat.element_size()
        for p in self._early:
            g = p.grad
            if g is None or not (lo <= g.data_ptr() < hi):
                return False
        return True

    def _arrived(self, _param):
        self._count += 1
        if not self._active():
            return
        if self._count > self._early_n and (self.early_launched_in_backward or self._reduced):
            raise RuntimeError("GradBuckets: a second backward pass reached the early parameters before the step ended "
                               "(optimizer.zero_grad() / begin()): their slice is already being summed over the ranks")
        if self._count == self._early_n and self._early_grads_in_flat():
            self._work.append(dist.all_reduce(self.flat[self.lo:self.hi], op=dist.ReduceOp.SUM, async_op=True))
            self.early_launched_in_backward = True

    def finish(self):
        """After backward: reduce the rest, wait for everything, average.  Returns the bytes this rank put on the wire."""
        if not self._active():
            return 0
        if self._reduced:
            raise RuntimeError("GradBuckets.finish(): this step's gradients were already averaged")
        if self._early_n and not self.early_launched_in_backward:          # e.g. a parameter that took no part in this step
            self._work.append(dist.all_reduce(self.flat[self.lo:self.hi], op=dist.ReduceOp.SUM, async_op=True))
        for a, b in ((0, self.lo), (self.hi, self.n)):
            if b > a:
                self._work.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
        for w in self._work:
            w.wait()
        self._work = []
        self.flat[:self.n].div_(dist.get_world_size())
        self._reduced = True
        return self.n * self.flat.element_size()

    def wait_early(self):
        """Block until the slice a backward hook put on the wire has arrived (before anything else writes into it)."""
        for w in self._work:
            w.wait()
        self._work = []


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
