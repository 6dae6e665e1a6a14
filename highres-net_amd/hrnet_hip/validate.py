"""Validation sharded over the ranks (last sentence; reference src/train.py:196-215).

The reference scores every validation imageset on rank 0: `srs = fusion_model(lrs, alphas)[:, 0]`, then per sample
`val_score -= shift_cPSNR(np.clip(srs[i], 0, 1), hrs[i], hr_maps[i])` on the host, finally `val_score /= len(dataset)`.  Here every rank
scores ITS imagesets on the device (`HRNet` in eval mode + `hrn_shift_cpsnr`, 49 shifted cPSNRs per image in one launch) and the two
scalars (sum of scores, number of samples) are all-reduced: one 16-byte collective per validation pass, no image ever leaves its GPU.
Without a process group the result is the single-process score.
"""
import torch
import torch.distributed as dist

from . import binding


def shard_indices(n_items, rank, world_size):
    """Imagesets of rank `rank`: every world_size-th one (ragged tails spread over the first ranks)."""
    return list(range(rank, n_items, world_size))


def sharded_val_score(fusion_model, batches, border_w=3, score_fn=None, device=None):
    """`batches`: this rank's validation batches of (lrs, alphas, hrs, hr_maps) tensors (the reference loads them one imageset at a
    time, train.py:281).  Returns -mean(shift_cPSNR) over ALL ranks' samples, as `val_score` of train.py:199-215.
    score_fn(srs (B,S,S), hrs, hr_maps) -> (B,) replaces `hrn_shift_cpsnr` in the CPU rehearsal of the collective (tests/test_dist_cpu.py)."""
    score_fn = score_fn or (lambda s, h, m: binding.shift_cpsnr(s, h, m, border_w, True))
    was_training = fusion_model.training
    fusion_model.eval()
    total, count = None, 0
    try:
        with torch.no_grad():
            for lrs, alphas, hrs, hr_maps in batches:
                srs = fusion_model(lrs, alphas)[:, 0]
                sc = score_fn(srs, hrs, hr_maps).double().sum()
                total = sc if total is None else total + sc
                count += int(srs.shape[0])
    finally:
        fusion_model.train(was_training)
    if total is None:
        total = torch.zeros((), dtype=torch.float64, device=device or "cpu")
    acc = torch.stack([total.reshape(()), torch.tensor(float(count), dtype=torch.float64, device=total.device)])
    if dist.is_initialized() and dist.get_world_size() > 1:
        if acc.device.type != "cuda" and dist.get_backend() == "nccl":
            acc = acc.cuda()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    if float(acc[1]) == 0:
        raise ValueError("sharded_val_score: no validation sample on any rank")
    return -float(acc[0] / acc[1])
