"""The registered-loss tail of a training step on device (reference src/train.py:66-106, :183-187).

`get_loss(srs, hrs, hr_maps, metric, crop)` has the reference's signature (train.py:66) plus `crop`, which folds
`get_crop_mask` (train.py:90-106) into the same pass: callers that multiply the mask themselves pass crop=0.  For 'cMSE' and
'cPSNR' with a gradient-requiring `srs` it is a `torch.autograd.Function` over `hrn_get_loss_train` / `hrn_get_loss_backward`:
ONE forward pass over the Lanczos output producing n, the brightness bias b and cMSE per sample, and ONE backward pass producing
d(srs) with b held constant - the reference detaches it (train.py:83).  Without gradients (validation) it is the forward-only
`hrn_get_loss`.  No PyTorch fallback."""
import torch

from . import binding


class _RegisteredLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, srs, hrs, hr_maps, metric, crop):
        out, stats = binding.get_loss_train(srs.detach(), hrs, hr_maps, metric, crop)
        ctx.save_for_backward(srs.detach(), hrs, hr_maps, stats)
        ctx.metric, ctx.crop = metric, crop
        return out

    @staticmethod
    def backward(ctx, d_out):
        srs, hrs, hr_maps, stats = ctx.saved_tensors
        d_srs = binding.get_loss_backward(srs, hrs, hr_maps, stats, d_out.contiguous(), ctx.metric, ctx.crop)
        return d_srs, None, None, None, None


def get_loss(srs, hrs, hr_maps, metric="cMSE", crop=0):
    """(B,S,S) x 3 -> (B,) 'masked_MSE' | 'cMSE' | 'cPSNR' (= -10 log10 cMSE, as the reference returns it)."""
    if torch.is_grad_enabled() and torch.is_tensor(srs) and srs.requires_grad:
        if metric == "masked_MSE":
            raise NotImplementedError("masked_MSE has no device backward (train.py only trains on 'cPSNR')")
        return _RegisteredLoss.apply(srs, hrs, hr_maps, metric, int(crop))
    return binding.get_loss(srs, hrs, hr_maps, metric, crop)
