"""lanczos on MI355X: the reference's `src/lanczos.py` module surface (lanczos_kernel :5-43, lanczos_shift :47-107)
backed by the fused gfx950 kernels `hrn_lanczos_kernel` / `hrn_lanczos_shift` (tap synthesis + separable 7x7
gather with a reflect halo in LDS, one launch for all images instead of a Python loop over channels)."""
import torch

from hrnet_hip import binding


def _only_lanczos3(a, N):
    if a != 3 or N != 7:
        raise NotImplementedError(f"the gfx950 kernel implements a=3, N=7 (every call site of the reference); got a={a}, N={N}")


def lanczos_kernel(dx, a=3, N=7, dtype=None, device=None):
    """1-D Lanczos taps.  dx: tensor (M, 1) of shifts (or a nested list) -> (M, N)."""
    _only_lanczos3(a, N)
    if not torch.is_tensor(dx):
        dx = torch.tensor(dx, dtype=dtype, device=device)
    if device is not None:
        dx = dx.to(device)
    out = binding.lanczos_kernel(dx)
    want = dtype if dtype is not None else dx.dtype
    return out if want == torch.float32 else out.to(want)


def lanczos_shift(img, shift, p=3, a=3, N=7):
    """Shift every channel c of img (b, c, H, W) by shift[c] = (dy, dx) sub-pixels -> (b, c, H, W)."""
    _only_lanczos3(a, N)
    if p < N // 2:
        raise ValueError(f"padding p={p} must cover the kernel radius {N // 2}")
    if min(img.shape[-2:]) <= p:
        raise ValueError("reflection padding needs p < H, W")       # same condition nn.ReflectionPad2d enforces
    if torch.is_grad_enabled() and ((torch.is_tensor(img) and img.requires_grad) or (torch.is_tensor(shift) and shift.requires_grad)):
        out = _LanczosShiftFunction.apply(img, shift)          # differentiable wrt the image and the shifts (apply_shifts)
    else:
        out = torch.ops.hrnet_hip.lanczos_shift(img, shift) if img.is_cuda and shift.is_cuda else binding.lanczos_shift(img, shift)
    return out if img.dtype == torch.float32 else out.to(img.dtype)


class _LanczosShiftFunction(torch.autograd.Function):
    """lanczos_shift with the HIP backward: d img = adjoint of the shift, d shift = gradient through the 7 taps per axis
    (the reference gets both from torch autograd over lanczos.py:5-107)."""

    @staticmethod
    def forward(ctx, img, shift):
        ctx.save_for_backward(img, shift)
        return binding.lanczos_shift(img, shift)

    @staticmethod
    def backward(ctx, d_out):
        img, shift = ctx.saved_tensors
        need_img, need_shift = ctx.needs_input_grad
        d_img, d_shift = binding.lanczos_shift_backward(img, shift, d_out.contiguous(), need_img, need_shift)
        if d_img is not None and d_img.dtype != img.dtype:
            d_img = d_img.to(img.dtype)
        if d_shift is not None:
            full = torch.zeros_like(shift, dtype=torch.float32)
            full[:d_shift.shape[0]] = d_shift                   # shift may carry more rows than img has channels
            d_shift = full.to(shift.dtype)
        return d_img, d_shift
