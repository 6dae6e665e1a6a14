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
    if torch.is_tensor(img) and torch.is_tensor(shift) and img.is_cuda and shift.is_cuda:
        # dispatcher-registered op, differentiable wrt the image and the shifts (apply_shifts, train.py:47-63): its autograd formula is
        # torch.ops.hrnet_hip.lanczos_shift_backward (adjoint of the shift + the gradient through the 7 taps per axis)
        out = torch.ops.hrnet_hip.lanczos_shift(img if img.dtype == torch.float32 else img.float(), shift if shift.dtype == torch.float32 else shift.float())
    else:
        out = binding.lanczos_shift(img, shift)                 # raises: no CPU fallback
    return out if img.dtype == torch.float32 else out.to(img.dtype)
