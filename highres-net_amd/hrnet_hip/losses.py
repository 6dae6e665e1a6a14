"""The registered-loss tail of a training step on device (reference src/train.py:66-106, :183-187).

`get_loss(srs, hrs, hr_maps, metric, crop)` has the reference's signature (train.py:66) plus `crop`, which folds
`get_crop_mask` (train.py:90-106) into the same pass: callers that multiply the mask themselves pass crop=0.  For 'cMSE' and
'cPSNR' with a gradient-requiring `srs` it is the dispatcher-registered op `torch.ops.hrnet_hip.get_loss_train` (autograd formula: `get_loss_backward`) over `hrn_get_loss_train` / `hrn_get_loss_backward`:
ONE forward pass over the Lanczos output producing n, the brightness bias b and cMSE per sample, and ONE backward pass producing
d(srs) with b held constant - the reference detaches it (train.py:83).  Without gradients (validation) it is the forward-only
`hrn_get_loss`.  No PyTorch fallback."""
import torch

from . import binding


def get_loss(srs, hrs, hr_maps, metric="cMSE", crop=0):
    """(B,S,S) x 3 -> (B,) 'masked_MSE' | 'cMSE' | 'cPSNR' (= -10 log10 cMSE, as the reference returns it)."""
    if torch.is_grad_enabled() and torch.is_tensor(srs) and srs.requires_grad:
        if metric == "masked_MSE":
            raise NotImplementedError("masked_MSE has no device backward (train.py only trains on 'cPSNR')")
        out, _stats = torch.ops.hrnet_hip.get_loss_train(srs if srs.dtype == torch.float32 else srs.float(), hrs.float(), hr_maps.float(), metric, int(crop))
        return out
    return binding.get_loss(srs, hrs, hr_maps, metric, crop)
