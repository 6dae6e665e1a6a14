"""ctypes binding of libhrnet_hip.so (C ABI: include/hrnet_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory (`tensor.data_ptr()`), the current HIP stream and, for
multi-GPU, `torch.distributed`.  All arithmetic happens in the hand-written gfx950 kernels of the library.
There is NO fallback: if the library is missing or a tensor is not on a ROCm device, these functions raise.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRNET_HIP_LIB") or os.path.join(_HERE, "libhrnet_hip.so")   # override: A/B-testing a build

F32, BF16, BF16X3 = 0, 1, 2
MAX_RES_LAYERS = 8
_DT_TORCH = {F32: torch.float32, BF16: torch.bfloat16, BF16X3: torch.bfloat16}     # BF16X3: two bf16 planes (hi, lo), planes first


def planes_to_float(t, dtype):
    """A stage tensor as float32: bf16x3 stage tensors are (2, ...) bf16 pairs of planes, value = hi + lo."""
    return t[0].float() + t[1].float() if dtype == BF16X3 else t.float()


def float_to_planes(x, dtype):
    """float32 -> the stage tensor of `dtype` (bf16x3: hi = bf16(x), lo = bf16(x - hi), stacked planes first)."""
    if dtype == BF16X3:
        hi = x.to(torch.bfloat16)
        return torch.stack([hi, (x - hi.float()).to(torch.bfloat16)]).contiguous()
    return x.to(_DT_TORCH[dtype]).contiguous()
_fp = ctypes.POINTER(ctypes.c_float)


class HrnetParams(ctypes.Structure):
    _fields_ = [
        ("num_layers", ctypes.c_int),
        ("enc_init_w", ctypes.c_void_p), ("enc_init_b", ctypes.c_void_p), ("enc_init_a", ctypes.c_void_p),
        ("enc_res_w", ctypes.c_void_p * (2 * MAX_RES_LAYERS)),
        ("enc_res_b", ctypes.c_void_p * (2 * MAX_RES_LAYERS)),
        ("enc_res_a", ctypes.c_void_p * (2 * MAX_RES_LAYERS)),
        ("enc_final_w", ctypes.c_void_p), ("enc_final_b", ctypes.c_void_p),
        ("fuse_res_w", ctypes.c_void_p * 2), ("fuse_res_b", ctypes.c_void_p * 2), ("fuse_res_a", ctypes.c_void_p * 2),
        ("fuse_out_w", ctypes.c_void_p), ("fuse_out_b", ctypes.c_void_p), ("fuse_out_a", ctypes.c_void_p),
        ("dec_w", ctypes.c_void_p), ("dec_b", ctypes.c_void_p), ("dec_a", ctypes.c_void_p),
        ("fin_w", ctypes.c_void_p), ("fin_b", ctypes.c_void_p),
    ]


class ShiftnetParams(ctypes.Structure):
    _fields_ = [
        ("conv_w", ctypes.c_void_p * 8), ("conv_b", ctypes.c_void_p * 8),
        ("bn_g", ctypes.c_void_p * 8), ("bn_b", ctypes.c_void_p * 8),
        ("bn_rm", ctypes.c_void_p * 8), ("bn_rv", ctypes.c_void_p * 8),
        ("fc1_w", ctypes.c_void_p), ("fc1_b", ctypes.c_void_p), ("fc2_w", ctypes.c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol include/hrnet_hip.h declares (checked by tests/test_abi.py)
_c = ctypes
SIGNATURES = {
    "hrn_version": (_c.c_int, []),
    "hrn_last_error": (_c.c_char_p, []),
    "hrn_hrnet_packed_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "hrn_hrnet_pack": (_c.c_int, [_c.POINTER(HrnetParams), _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_hrnet_workspace_bytes": (_c.c_size_t, [_c.c_int] * 5),
    "hrn_hrnet_forward": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                     _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_encoder_forward": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_fuse_forward": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int,
                                    _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_decoder_forward": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_void_p, _c.c_void_p]),
    "hrn_hrnet_train_workspace_bytes": (_c.c_size_t, [_c.c_int] * 5),
    "hrn_hrnet_forward_train": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_hrnet_backward": (_c.c_int, [_c.c_void_p, _c.POINTER(HrnetParams), _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int,
                                      _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(HrnetParams), _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_hrnet_forward_train_dt": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                              _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_hrnet_backward_dt": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(HrnetParams), _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int,
                                         _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(HrnetParams), _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_shiftnet_packed_bytes": (_c.c_size_t, []),
    "hrn_shiftnet_pack": (_c.c_int, [_c.POINTER(ShiftnetParams), _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_shiftnet_workspace_bytes": (_c.c_size_t, [_c.c_int]),
    "hrn_shiftnet_forward": (_c.c_int, [_c.c_void_p, _c.POINTER(ShiftnetParams), _c.c_void_p, _c.c_int, _c.c_int, _c.c_float,
                                        _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_shiftnet_train_workspace_bytes": (_c.c_size_t, [_c.c_int]),
    "hrn_shiftnet_forward_train": (_c.c_int, [_c.c_void_p, _c.POINTER(ShiftnetParams), _c.c_void_p, _c.c_int, _c.c_float, _c.c_void_p,
                                              _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_shiftnet_backward": (_c.c_int, [_c.POINTER(ShiftnetParams), _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                         _c.POINTER(ShiftnetParams), _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_adam_step": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_float, _c.c_float, _c.c_float,
                                 _c.c_float, _c.c_float, _c.c_int, _c.c_void_p]),
    "hrn_lanczos_kernel": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "hrn_lanczos_shift": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "hrn_lanczos_shift_backward_workspace_bytes": (_c.c_size_t, [_c.c_int] * 4),
    "hrn_lanczos_shift_backward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                              _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_get_loss": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "hrn_get_loss_train_workspace_bytes": (_c.c_size_t, [_c.c_int]),
    "hrn_get_loss_train": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                      _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_get_loss_backward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                         _c.c_int, _c.c_void_p, _c.c_void_p]),
    "hrn_shift_cpsnr_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "hrn_shift_cpsnr": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                   _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "hrn_profile_enable": (_c.c_int, [_c.c_int]),
    "hrn_profile_count": (_c.c_int, []),
    "hrn_profile_get": (_c.c_int, [_c.c_int, _c.c_char_p, _c.c_int, _c.POINTER(_c.c_long), _c.POINTER(_c.c_double),
                                   _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
}

_lib = None
_lock = threading.Lock()


def load_library():
    """dlopen the in-tree library and type every entry point.  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python highres-net_amd/hrnet_hip/build.py` "
                "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback for the HIP path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here == ABI mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class HrnetHipError(RuntimeError):
    pass


def has_bf16x3():
    """True when the loaded library implements the split-bf16 precision mode (HRN_DTYPE_BF16X3)."""
    return load_library().hrn_hrnet_packed_bytes(BF16X3, 2) != 0


def _check(rc, what):
    if rc != 0:
        msg = load_library().hrn_last_error().decode("utf-8", "replace")
        raise HrnetHipError(f"{what} failed with code {rc}: {msg}")


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on '{t.device}': the HIP path needs ROCm device tensors (no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- HRNet
def hrnet_param_struct(named, num_layers):
    """named: dict of reference state-dict keys -> device f32 tensors.  Returns (HrnetParams, tensors kept alive)."""
    keep = []

    def p(key):
        t = _dev_f32(named[key].detach(), key)
        keep.append(t)
        return t.data_ptr()

    P = HrnetParams()
    P.num_layers = num_layers
    P.enc_init_w, P.enc_init_b, P.enc_init_a = p("encode.init_layer.0.weight"), p("encode.init_layer.0.bias"), p("encode.init_layer.1.weight")
    for l in range(num_layers):
        for j, (cw, ca) in enumerate(((0, 1), (2, 3))):
            P.enc_res_w[2 * l + j] = p(f"encode.res_layers.{l}.block.{cw}.weight")
            P.enc_res_b[2 * l + j] = p(f"encode.res_layers.{l}.block.{cw}.bias")
            P.enc_res_a[2 * l + j] = p(f"encode.res_layers.{l}.block.{ca}.weight")
    P.enc_final_w, P.enc_final_b = p("encode.final.0.weight"), p("encode.final.0.bias")
    for j, (cw, ca) in enumerate(((0, 1), (2, 3))):
        P.fuse_res_w[j] = p(f"fuse.fuse.0.block.{cw}.weight")
        P.fuse_res_b[j] = p(f"fuse.fuse.0.block.{cw}.bias")
        P.fuse_res_a[j] = p(f"fuse.fuse.0.block.{ca}.weight")
    P.fuse_out_w, P.fuse_out_b, P.fuse_out_a = p("fuse.fuse.1.weight"), p("fuse.fuse.1.bias"), p("fuse.fuse.2.weight")
    P.dec_w, P.dec_b, P.dec_a = p("decode.deconv.0.weight"), p("decode.deconv.0.bias"), p("decode.deconv.1.weight")
    P.fin_w, P.fin_b = p("decode.final.weight"), p("decode.final.bias")
    return P, keep


def hrnet_pack(named, num_layers, dtype):
    """named: dict of reference state-dict keys -> device f32 tensors.  Returns the packed uint8 tensor."""
    lib = load_library()
    P, keep = hrnet_param_struct(named, num_layers)
    nbytes = lib.hrn_hrnet_packed_bytes(dtype, num_layers)
    if nbytes == 0:
        raise HrnetHipError(f"unsupported dtype/num_layers ({dtype}, {num_layers})")
    dev = keep[0].device
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib.hrn_hrnet_pack(ctypes.byref(P), dtype, _ptr(packed), nbytes, _stream()), "hrn_hrnet_pack")
    return packed


_ws_cache = {}


def _workspace(nbytes, device, tag):
    key = (tag, device.index)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _ws_cache.pop(key, None)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def hrnet_workspace(dtype, B, V, H, W, device):
    n = load_library().hrn_hrnet_workspace_bytes(dtype, B, V, H, W)
    if n == 0:
        raise HrnetHipError(f"bad HRNet problem size B={B} V={V} H={H} W={W}")
    return _workspace(n, device, "hrnet")


def hrnet_forward(packed, dtype, num_layers, alpha_residual, lrs, alphas, out=None):
    lib = load_library()
    lrs = _dev_f32(lrs, "lrs")
    alphas = _dev_f32(alphas, "alphas").to(lrs.device)
    if lrs.dim() != 4 or alphas.shape != lrs.shape[:2]:
        raise ValueError(f"lrs must be (B,V,H,W) and alphas (B,V); got {tuple(lrs.shape)} / {tuple(alphas.shape)}")
    B, V, H, W = lrs.shape
    with torch.cuda.device(lrs.device):
        ws = hrnet_workspace(dtype, B, V, H, W, lrs.device)
        sr = out if out is not None else torch.empty((B, 1, 3 * H, 3 * W), dtype=torch.float32, device=lrs.device)
        _check(lib.hrn_hrnet_forward(_ptr(packed), dtype, num_layers, int(bool(alpha_residual)), _ptr(lrs), _ptr(alphas),
                                     B, V, H, W, _ptr(sr), _ptr(ws), ws.numel(), _stream()), "hrn_hrnet_forward")
    return sr


def hrnet_encoder(packed, dtype, num_layers, lrs):
    """-> view stack (B,V,H,W,64) channels-last in the storage dtype ((2,B,V,H,W,64) bf16 planes for BF16X3)."""
    lib = load_library()
    lrs = _dev_f32(lrs, "lrs")
    B, V, H, W = lrs.shape
    with torch.cuda.device(lrs.device):
        ws = hrnet_workspace(dtype, B, V, H, W, lrs.device)
        emb = torch.empty(((2,) if dtype == BF16X3 else ()) + (B, V, H, W, 64), dtype=_DT_TORCH[dtype], device=lrs.device)
        _check(lib.hrn_encoder_forward(_ptr(packed), dtype, num_layers, _ptr(lrs), B, V, H, W, _ptr(emb), _ptr(ws), ws.numel(),
                                       _stream()), "hrn_encoder_forward")
    return emb


def hrnet_fuse(packed, dtype, num_layers, alpha_residual, emb, alphas):
    """emb (B,V,H,W,64) storage dtype (destroyed) -> fused (B,H,W,64); BF16X3: both with a leading plane axis of 2."""
    lib = load_library()
    if emb.dtype != _DT_TORCH[dtype] or not emb.is_contiguous() or not emb.is_cuda or emb.dim() != (6 if dtype == BF16X3 else 5):
        raise ValueError("emb must be a contiguous device tensor in the storage dtype")
    B, V, H, W, _ = emb.shape[-5:]
    alphas = _dev_f32(alphas, "alphas")
    with torch.cuda.device(emb.device):
        ws = hrnet_workspace(dtype, B, V, H, W, emb.device)
        fused = torch.empty(((2,) if dtype == BF16X3 else ()) + (B, H, W, 64), dtype=emb.dtype, device=emb.device)
        _check(lib.hrn_fuse_forward(_ptr(packed), dtype, num_layers, int(bool(alpha_residual)), _ptr(emb), _ptr(alphas),
                                    B, V, H, W, _ptr(fused), _ptr(ws), ws.numel(), _stream()), "hrn_fuse_forward")
    return fused


def hrnet_decoder(packed, dtype, num_layers, fused):
    lib = load_library()
    if fused.dtype != _DT_TORCH[dtype] or not fused.is_contiguous() or not fused.is_cuda or fused.dim() != (5 if dtype == BF16X3 else 4):
        raise ValueError("fused must be a contiguous device tensor in the storage dtype")
    N, H, W, _ = fused.shape[-4:]
    with torch.cuda.device(fused.device):
        sr = torch.empty((N, 1, 3 * H, 3 * W), dtype=torch.float32, device=fused.device)
        _check(lib.hrn_decoder_forward(_ptr(packed), dtype, num_layers, _ptr(fused), N, H, W, _ptr(sr), _stream()), "hrn_decoder_forward")
    return sr


def hrnet_forward_train(packed_f32, lrs, alphas, num_layers, alpha_residual, dtype=F32):
    """Training forward: returns (sr, train_ws); train_ws holds every intermediate for hrnet_backward.  dtype F32 (exact-fp32 MFMA) or
    BF16X3 (split-bf16: `packed_f32` is then the BF16X3 blob and the workspace holds pairs of bf16 planes)."""
    lib = load_library()
    lrs = _dev_f32(lrs, "lrs")
    alphas = _dev_f32(alphas, "alphas")
    B, V, H, W = lrs.shape
    nbytes = lib.hrn_hrnet_train_workspace_bytes(num_layers, B, V, H, W)
    if nbytes == 0:
        raise HrnetHipError(f"bad training shape B={B} V={V} H={H} W={W} num_layers={num_layers}")
    tws = torch.empty(nbytes, dtype=torch.uint8, device=lrs.device)
    sr = torch.empty((B, 1, 3 * H, 3 * W), dtype=torch.float32, device=lrs.device)
    with torch.cuda.device(lrs.device):
        _check(lib.hrn_hrnet_forward_train_dt(_ptr(packed_f32), int(dtype), num_layers, int(bool(alpha_residual)), _ptr(lrs), _ptr(alphas),
                                              B, V, H, W, _ptr(sr), _ptr(tws), nbytes, _stream()), "hrn_hrnet_forward_train")
    return sr, tws


def hrnet_backward(packed_f32, named_params, named_grads, num_layers, alpha_residual, lrs, alphas, d_sr, tws, dtype=F32):
    """Accumulates dLoss/dparam into named_grads (same keys / shapes as named_params, f32, zero them for plain gradients)."""
    lib = load_library()
    lrs = _dev_f32(lrs, "lrs")
    alphas = _dev_f32(alphas, "alphas")
    d_sr = _dev_f32(d_sr, "d_sr")
    B, V, H, W = lrs.shape
    if tuple(d_sr.shape) != (B, 1, 3 * H, 3 * W):
        raise ValueError(f"d_sr shape {tuple(d_sr.shape)} != {(B, 1, 3 * H, 3 * W)}")
    P, keep_p = hrnet_param_struct(named_params, num_layers)
    G, keep_g = hrnet_param_struct(named_grads, num_layers)
    for t, g in zip(keep_p, keep_g):
        if t.shape != g.shape or g.data_ptr() == t.data_ptr():
            raise ValueError("gradient buffers must match the parameters' shapes and not alias them")
    with torch.cuda.device(lrs.device):
        _check(lib.hrn_hrnet_backward_dt(_ptr(packed_f32), int(dtype), ctypes.byref(P), int(bool(alpha_residual)), _ptr(lrs), _ptr(alphas),
                                         B, V, H, W, _ptr(d_sr), ctypes.byref(G), _ptr(tws), tws.numel(), _stream()),
               "hrn_hrnet_backward")


# --------------------------------------------------------------------------- ShiftNet
def _shiftnet_struct(named, keep, with_weights):
    P = ShiftnetParams()

    def p(key):
        t = named[key].detach()
        if not t.is_cuda:
            raise RuntimeError(f"{key} is on '{t.device}': ShiftNet parameters must live on the ROCm device")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"{key} must be contiguous float32")
        keep.append(t)
        return t.data_ptr()

    for i in range(8):
        if with_weights:
            P.conv_w[i], P.conv_b[i] = p(f"layer{i + 1}.0.weight"), p(f"layer{i + 1}.0.bias")
        P.bn_g[i], P.bn_b[i] = p(f"layer{i + 1}.1.weight"), p(f"layer{i + 1}.1.bias")
        P.bn_rm[i], P.bn_rv[i] = p(f"layer{i + 1}.1.running_mean"), p(f"layer{i + 1}.1.running_var")
    if with_weights:
        P.fc1_w, P.fc1_b, P.fc2_w = p("fc1.weight"), p("fc1.bias"), p("fc2.weight")
    return P


def shiftnet_pack(named):
    lib = load_library()
    keep = []
    P = _shiftnet_struct(named, keep, True)
    dev = keep[0].device
    nbytes = lib.hrn_shiftnet_packed_bytes()
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib.hrn_shiftnet_pack(ctypes.byref(P), _ptr(packed), nbytes, _stream()), "hrn_shiftnet_pack")
    return packed


def shiftnet_forward(packed, named, x, train_bn=False, momentum=0.1, dropout_mask=None):
    """x (B,2,128,128) -> theta (B,2).  `named` supplies the live BatchNorm tensors (running stats are updated in
    place when train_bn) and fc1.weight, which the kernel reads in place.  dropout_mask: None or uint8 (B,32768) keep-mask in the reference's flatten order."""
    lib = load_library()
    x = _dev_f32(x, "x")
    if x.dim() != 4 or tuple(x.shape[1:]) != (2, 128, 128):
        raise ValueError(f"ShiftNet input must be (B,2,128,128) (fc1 is hard-wired to 128*16*16, ShiftNet.py:44); got {tuple(x.shape)}")
    B = x.shape[0]
    keep = []
    P = _shiftnet_struct(named, keep, True)        # (fc1.weight is read in place by the kernel: not part of `packed`)
    mptr = ctypes.c_void_p(0)
    if dropout_mask is not None:
        if dropout_mask.dtype != torch.uint8 or tuple(dropout_mask.shape) != (B, 32768) or not dropout_mask.is_cuda:
            raise ValueError("dropout_mask must be a uint8 device tensor of shape (B, 32768)")
        dropout_mask = dropout_mask.contiguous()
        mptr = _ptr(dropout_mask)
    with torch.cuda.device(x.device):
        nws = lib.hrn_shiftnet_workspace_bytes(B)
        ws = _workspace(nws, x.device, "shiftnet")
        theta = torch.empty((B, 2), dtype=torch.float32, device=x.device)
        _check(lib.hrn_shiftnet_forward(_ptr(packed), ctypes.byref(P), _ptr(x), B, int(bool(train_bn)), float(momentum), mptr,
                                        _ptr(theta), _ptr(ws), ws.numel(), _stream()), "hrn_shiftnet_forward")
    return theta


def _check_shiftnet_input(x, dropout_mask):
    if x.dim() != 4 or tuple(x.shape[1:]) != (2, 128, 128):
        raise ValueError(f"ShiftNet input must be (B,2,128,128) (fc1 is hard-wired to 128*16*16, ShiftNet.py:44); got {tuple(x.shape)}")
    B = x.shape[0]
    if dropout_mask is None:
        return ctypes.c_void_p(0), None
    if dropout_mask.dtype != torch.uint8 or tuple(dropout_mask.shape) != (B, 32768) or not dropout_mask.is_cuda:
        raise ValueError("dropout_mask must be a uint8 device tensor of shape (B, 32768)")
    dropout_mask = dropout_mask.contiguous()
    return _ptr(dropout_mask), dropout_mask


def shiftnet_forward_train(packed, named, x, momentum=0.1, dropout_mask=None):
    """Train-mode forward that keeps its intermediates: returns (theta (B,2), train_ws)."""
    lib = load_library()
    x = _dev_f32(x, "x")
    mptr, dropout_mask = _check_shiftnet_input(x, dropout_mask)
    B = x.shape[0]
    keep = []
    P = _shiftnet_struct(named, keep, True)        # (fc1.weight is read in place by the kernel: not part of `packed`)
    nbytes = lib.hrn_shiftnet_train_workspace_bytes(B)
    tws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    theta = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.hrn_shiftnet_forward_train(_ptr(packed), ctypes.byref(P), _ptr(x), B, float(momentum), mptr, _ptr(theta),
                                              _ptr(tws), nbytes, _stream()), "hrn_shiftnet_forward_train")
    return theta, tws


def shiftnet_backward(named, named_grads, x, dropout_mask, d_theta, tws, need_input_grad=True):
    """Accumulates the parameter gradients into named_grads (parameter keys only); returns d_x (B,2,128,128) or None."""
    lib = load_library()
    x = _dev_f32(x, "x")
    d_theta = _dev_f32(d_theta, "d_theta")
    mptr, dropout_mask = _check_shiftnet_input(x, dropout_mask)
    B = x.shape[0]
    keep = []
    P = _shiftnet_struct(named, keep, True)
    gfull = dict(named_grads)
    for k, v in named.items():                               # the struct builder also wants the (unused) running stats
        gfull.setdefault(k, v)
    G = _shiftnet_struct(gfull, keep, True)
    d_x = torch.empty_like(x) if need_input_grad else None
    with torch.cuda.device(x.device):
        _check(lib.hrn_shiftnet_backward(ctypes.byref(P), _ptr(x), B, mptr, _ptr(d_theta), ctypes.byref(G),
                                         _ptr(d_x) if need_input_grad else None, _ptr(tws), tws.numel(), _stream()),
               "hrn_shiftnet_backward")
    return d_x


# --------------------------------------------------------------------------- Lanczos
def lanczos_kernel(dx):
    lib = load_library()
    dx = _dev_f32(dx, "dx").reshape(-1)
    n = dx.numel()
    taps = torch.empty((n, 7), dtype=torch.float32, device=dx.device)
    with torch.cuda.device(dx.device):
        _check(lib.hrn_lanczos_kernel(_ptr(dx), n, _ptr(taps), _stream()), "hrn_lanczos_kernel")
    return taps


def lanczos_shift(img, shift):
    lib = load_library()
    img = _dev_f32(img, "img")
    shift = _dev_f32(shift, "shift").to(img.device)
    if img.dim() != 4 or shift.dim() != 2 or shift.shape[1] != 2 or shift.shape[0] < img.shape[1]:
        raise ValueError(f"img must be (b,c,H,W) and shift (c,2); got {tuple(img.shape)} / {tuple(shift.shape)}")
    b, c, H, W = img.shape
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _check(lib.hrn_lanczos_shift(_ptr(img), _ptr(shift), b, c, H, W, _ptr(out), _stream()), "hrn_lanczos_shift")
    return out


def lanczos_shift_backward(img, shift, d_out, need_img=True, need_shift=True):
    """Gradients of lanczos_shift(img, shift) given d_out: (d_img or None, d_shift (c, 2) or None)."""
    lib = load_library()
    img, shift, d_out = _dev_f32(img, "img"), _dev_f32(shift, "shift"), _dev_f32(d_out, "d_out")
    b, c, H, W = img.shape
    nbytes = lib.hrn_lanczos_shift_backward_workspace_bytes(b, c, H, W)
    ws = _workspace(nbytes, img.device, "lanczos_bwd")
    d_img = torch.empty_like(img) if need_img else None
    d_shift = torch.zeros((c, 2), dtype=torch.float32, device=img.device) if need_shift else None
    with torch.cuda.device(img.device):
        _check(lib.hrn_lanczos_shift_backward(_ptr(img), _ptr(shift), _ptr(d_out), b, c, H, W,
                                              _ptr(d_img) if need_img else None, _ptr(d_shift) if need_shift else None,
                                              _ptr(ws), ws.numel(), _stream()), "hrn_lanczos_shift_backward")
    return d_img, d_shift


# --------------------------------------------------------------------------- optimiser
# Bumped by anything that rewrites parameter storage without going through torch's version counters (the fused Adam
# kernel writes the flat buffer the parameters are views of); the modules' packed-parameter caches key on it.
param_epoch = 0


def bump_param_epoch():
    global param_epoch
    param_epoch += 1


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    """In-place Adam update of the flat fp32 device buffer `params` (torch.optim.Adam arithmetic, no amsgrad)."""
    lib = load_library()
    for name, t in (("params", params), ("grads", grads), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == params.numel()):
            raise ValueError(f"{name} must be a contiguous float32 device tensor of {params.numel()} elements")
    with torch.cuda.device(params.device):
        _check(lib.hrn_adam_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), float(lr), float(beta1),
                                 float(beta2), float(eps), float(weight_decay), int(step), _stream()), "hrn_adam_step")
    bump_param_epoch()


# --------------------------------------------------------------------------- built-in kernel timing
def profile_enable(on):
    _check(load_library().hrn_profile_enable(int(bool(on))), "hrn_profile_enable")


def profile_read():
    """-> {family: dict(launches, ms, flops, bytes)} for everything recorded since profile_enable(True)."""
    lib = load_library()
    out = {}
    for i in range(lib.hrn_profile_count()):
        name = ctypes.create_string_buffer(64)
        n, ms, fl, by = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(lib.hrn_profile_get(i, name, 64, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), "hrn_profile_get")
        out[name.value.decode()] = {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}
    return out


# --------------------------------------------------------------------------- loss / score reductions
_METRICS = {"masked_MSE": 0, "cMSE": 1, "cPSNR": 2}


def get_loss(srs, hrs, hr_maps, metric="cMSE", crop=0):
    """get_loss of the reference (train.py:66-87) on device: (B,S,S) tensors -> (B,).  `crop` folds get_crop_mask in."""
    lib = load_library()
    if metric not in _METRICS:
        raise ValueError(f"metric must be one of {sorted(_METRICS)}; got {metric!r}")
    srs, hrs, hr_maps = _dev_f32(srs, "srs"), _dev_f32(hrs, "hrs"), _dev_f32(hr_maps, "hr_maps")
    if srs.dim() != 3 or srs.shape != hrs.shape or srs.shape != hr_maps.shape or srs.shape[1] != srs.shape[2]:
        raise ValueError(f"srs, hrs, hr_maps must be equal (B,S,S) tensors; got {tuple(srs.shape)}, {tuple(hrs.shape)}, {tuple(hr_maps.shape)}")
    B, S, _ = srs.shape
    out = torch.empty((B,), dtype=torch.float32, device=srs.device)
    with torch.cuda.device(srs.device):
        _check(lib.hrn_get_loss(_ptr(srs), _ptr(hrs), _ptr(hr_maps), B, S, int(crop), _METRICS[metric], _ptr(out), _stream()), "hrn_get_loss")
    return out


def get_loss_train(srs, hrs, hr_maps, metric="cPSNR", crop=0):
    """Forward of the differentiable loss tail: -> (out (B,), stats (B,4) f64 = {n, bias, cMSE, 0})."""
    lib = load_library()
    if metric not in ("cMSE", "cPSNR"):
        raise ValueError(f"the registered loss is defined for 'cMSE' and 'cPSNR'; got {metric!r}")
    srs, hrs, hr_maps = _dev_f32(srs, "srs"), _dev_f32(hrs, "hrs"), _dev_f32(hr_maps, "hr_maps")
    if srs.dim() != 3 or srs.shape != hrs.shape or srs.shape != hr_maps.shape or srs.shape[1] != srs.shape[2]:
        raise ValueError(f"srs, hrs, hr_maps must be equal (B,S,S) tensors; got {tuple(srs.shape)}, {tuple(hrs.shape)}, {tuple(hr_maps.shape)}")
    B, S, _ = srs.shape
    out = torch.empty((B,), dtype=torch.float32, device=srs.device)
    stats = torch.empty((B, 4), dtype=torch.float64, device=srs.device)
    with torch.cuda.device(srs.device):
        ws = _workspace(lib.hrn_get_loss_train_workspace_bytes(B), srs.device, "loss_train")
        _check(lib.hrn_get_loss_train(_ptr(srs), _ptr(hrs), _ptr(hr_maps), B, S, int(crop), _METRICS[metric], _ptr(out), _ptr(stats),
                                      _ptr(ws), ws.numel(), _stream()), "hrn_get_loss_train")
    return out, stats


def get_loss_backward(srs, hrs, hr_maps, stats, d_out, metric="cPSNR", crop=0):
    """d_out (B,) -> d_srs (B,S,S): the brightness bias is a constant, as in the reference (train.py:83)."""
    lib = load_library()
    srs, hrs, hr_maps, d_out = _dev_f32(srs, "srs"), _dev_f32(hrs, "hrs"), _dev_f32(hr_maps, "hr_maps"), _dev_f32(d_out, "d_out")
    B, S, _ = srs.shape
    d_srs = torch.empty_like(srs)
    with torch.cuda.device(srs.device):
        _check(lib.hrn_get_loss_backward(_ptr(srs), _ptr(hrs), _ptr(hr_maps), _ptr(stats), _ptr(d_out), B, S, int(crop),
                                         _METRICS[metric], _ptr(d_srs), _stream()), "hrn_get_loss_backward")
    return d_srs


def shift_cpsnr(srs, hrs, hr_maps, border_w=3, clip=True):
    """Batched shift_cPSNR (Evaluator.py:52-73) on device: (B,S,S) tensors -> (B,) best cPSNR over the (2w+1)^2 offsets."""
    lib = load_library()
    srs, hrs, hr_maps = _dev_f32(srs, "srs"), _dev_f32(hrs, "hrs"), _dev_f32(hr_maps, "hr_maps")
    if srs.dim() == 2:
        srs, hrs, hr_maps = srs[None], hrs[None], hr_maps[None]
    if srs.dim() != 3 or srs.shape != hrs.shape or srs.shape != hr_maps.shape or srs.shape[1] != srs.shape[2]:
        raise ValueError("srs, hrs, hr_maps must be equal (B,S,S) tensors")
    B, S, _ = srs.shape
    nws = lib.hrn_shift_cpsnr_workspace_bytes(B, int(border_w))
    out = torch.empty((B,), dtype=torch.float32, device=srs.device)
    with torch.cuda.device(srs.device):
        ws = _workspace(nws, srs.device, "shift_cpsnr")
        _check(lib.hrn_shift_cpsnr(_ptr(srs.contiguous()), _ptr(hrs.contiguous()), _ptr(hr_maps.contiguous()), B, S, int(border_w),
                                   int(bool(clip)), _ptr(out), _ptr(ws), ws.numel(), _stream()), "hrn_shift_cpsnr")
    return out


# --------------------------------------------------------------------------- PyTorch-ROCm custom ops (north_star: "exposed to Python as
# PyTorch-ROCm custom ops"): the inference entry points are registered with the dispatcher as torch.ops.hrnet_hip.*, with fake
# (meta) implementations, so that they are visible to torch.compile / export and to anyone calling through torch.ops.  Each is a
# thin shim over the ctypes call above - the C ABI stays the boundary.  The reference-named modules call THESE in eval mode.
@torch.library.custom_op("hrnet_hip::hrnet_forward", mutates_args=(), device_types="cuda")
def _op_hrnet_forward(packed: torch.Tensor, dtype: int, num_layers: int, alpha_residual: bool, lrs: torch.Tensor,
                      alphas: torch.Tensor) -> torch.Tensor:
    return hrnet_forward(packed, dtype, num_layers, alpha_residual, lrs, alphas)


@_op_hrnet_forward.register_fake
def _(packed, dtype, num_layers, alpha_residual, lrs, alphas):
    b, _, h, w = lrs.shape
    return lrs.new_empty((b, 1, 3 * h, 3 * w), dtype=torch.float32)


@torch.library.custom_op("hrnet_hip::lanczos_shift", mutates_args=(), device_types="cuda")
def _op_lanczos_shift(img: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    return lanczos_shift(img, shift)


@_op_lanczos_shift.register_fake
def _(img, shift):
    return img.new_empty(img.shape, dtype=torch.float32)


@torch.library.custom_op("hrnet_hip::lanczos_kernel", mutates_args=(), device_types="cuda")
def _op_lanczos_kernel(dx: torch.Tensor) -> torch.Tensor:
    return lanczos_kernel(dx)


@_op_lanczos_kernel.register_fake
def _(dx):
    return dx.new_empty((dx.numel(), 7), dtype=torch.float32)


@torch.library.custom_op("hrnet_hip::shift_cpsnr", mutates_args=(), device_types="cuda")
def _op_shift_cpsnr(srs: torch.Tensor, hrs: torch.Tensor, hr_maps: torch.Tensor, border_w: int, clip: bool) -> torch.Tensor:
    return shift_cpsnr(srs, hrs, hr_maps, border_w, clip)


@_op_shift_cpsnr.register_fake
def _(srs, hrs, hr_maps, border_w, clip):
    return srs.new_empty((srs.shape[0] if srs.dim() == 3 else 1,), dtype=torch.float32)


# --------------------------------------------------------------------------- the TRAINING entry points as dispatcher-registered ops
# (call sites: src/train.py:174-191).  Each is registered with a fake (meta) implementation and, where the reference differentiates
# through it, with `register_autograd`: the backward formula is itself a registered op over the C ABI's *_backward entry point.  The
# reference-named modules call these through torch.ops.hrnet_hip.* in .train() mode.
from typing import List, Optional, Sequence, Tuple  # noqa: E402


def hrnet_param_names(num_layers):
    """HRNet's parameters in the module's registration order (== the reference's state_dict order, HRNet.py:36-169)."""
    names = ["encode.init_layer.0.weight", "encode.init_layer.0.bias", "encode.init_layer.1.weight"]
    for l in range(num_layers):
        names += [f"encode.res_layers.{l}.block.0.weight", f"encode.res_layers.{l}.block.0.bias", f"encode.res_layers.{l}.block.1.weight",
                  f"encode.res_layers.{l}.block.2.weight", f"encode.res_layers.{l}.block.2.bias", f"encode.res_layers.{l}.block.3.weight"]
    names += ["encode.final.0.weight", "encode.final.0.bias"]
    names += ["fuse.fuse.0.block.0.weight", "fuse.fuse.0.block.0.bias", "fuse.fuse.0.block.1.weight",
              "fuse.fuse.0.block.2.weight", "fuse.fuse.0.block.2.bias", "fuse.fuse.0.block.3.weight",
              "fuse.fuse.1.weight", "fuse.fuse.1.bias", "fuse.fuse.2.weight"]
    names += ["decode.deconv.0.weight", "decode.deconv.0.bias", "decode.deconv.1.weight", "decode.final.weight", "decode.final.bias"]
    return names


SHIFTNET_PARAM_NAMES = [f"layer{i}.{j}.{k}" for i in range(1, 9) for j in (0, 1) for k in ("weight", "bias")] + ["fc1.weight", "fc1.bias", "fc2.weight"]
SHIFTNET_BUFFER_NAMES = [f"layer{i}.1.{k}" for i in range(1, 9) for k in ("running_mean", "running_var")]


@torch.library.custom_op("hrnet_hip::hrnet_forward_train", mutates_args=(), device_types="cuda")
def _op_hrnet_forward_train(packed: torch.Tensor, lrs: torch.Tensor, alphas: torch.Tensor, params: Sequence[torch.Tensor],
                            num_layers: int, alpha_residual: bool, dtype: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`srs = fusion_model(lrs, alphas)` in training (train.py:174): the forward that keeps every intermediate in `tws`, in fp32 (dtype 0)
    or split-bf16 (dtype 2).  `packed` is the blob of `params` for that dtype (the raw parameters travel along for the backward pass and
    as the differentiable inputs)."""
    return hrnet_forward_train(packed, lrs, alphas, num_layers, alpha_residual, dtype)


@_op_hrnet_forward_train.register_fake
def _(packed, lrs, alphas, params, num_layers, alpha_residual, dtype):
    b, v, h, w = lrs.shape
    nbytes = load_library().hrn_hrnet_train_workspace_bytes(num_layers, b, v, h, w)
    return lrs.new_empty((b, 1, 3 * h, 3 * w), dtype=torch.float32), lrs.new_empty((nbytes,), dtype=torch.uint8)


@torch.library.custom_op("hrnet_hip::hrnet_backward", mutates_args=("tws",), device_types="cuda")     # (tws also holds the backward's scratch buffers)
def _op_hrnet_backward(packed: torch.Tensor, params: Sequence[torch.Tensor], lrs: torch.Tensor, alphas: torch.Tensor, d_sr: torch.Tensor,
                       tws: torch.Tensor, num_layers: int, alpha_residual: bool, dtype: int) -> List[torch.Tensor]:
    """d_sr -> the gradient of every parameter (train.py:190 through HRNet), in `hrnet_param_names` order."""
    names = hrnet_param_names(num_layers)
    named = dict(zip(names, params))
    grads = {k: torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for k, p in named.items()}
    hrnet_backward(packed, named, grads, num_layers, alpha_residual, lrs, alphas, d_sr.contiguous(), tws, dtype)
    return [grads[k] for k in names]


@_op_hrnet_backward.register_fake
def _(packed, params, lrs, alphas, d_sr, tws, num_layers, alpha_residual, dtype):
    return [p.new_empty(p.shape, dtype=torch.float32) for p in params]


def _hrnet_train_setup(ctx, inputs, output):
    packed, lrs, alphas, params, num_layers, alpha_residual, dtype = inputs
    ctx.num_layers, ctx.alpha_residual, ctx.n, ctx.dtype = num_layers, alpha_residual, len(params), dtype
    ctx.set_materialize_grads(False)          # (or autograd hands the backward a zero-filled "gradient" of the 20 GB workspace output)
    ctx.save_for_backward(packed, lrs, alphas, output[1], *params)


def _hrnet_train_backward(ctx, d_sr, _d_tws):
    packed, lrs, alphas, tws, *params = ctx.saved_tensors
    if d_sr is None:
        return None, None, None, [None] * len(params), None, None, None
    # (tws.data: the backward's scratch buffers live in tws too, so the op declares it mutated; through an alias with its own version
    # counter the saved tensor stays valid for a second backward pass - backward(retain_graph=True), the kept intermediates are only read)
    grads = torch.ops.hrnet_hip.hrnet_backward(packed, params, lrs, alphas, d_sr, tws.data, ctx.num_layers, ctx.alpha_residual, ctx.dtype)
    return None, None, None, [g.to(p.dtype) for g, p in zip(grads, params)], None, None, None


_op_hrnet_forward_train.register_autograd(_hrnet_train_backward, setup_context=_hrnet_train_setup)


def _shiftnet_named(params, buffers):
    named = dict(zip(SHIFTNET_PARAM_NAMES, params))
    named.update(zip(SHIFTNET_BUFFER_NAMES, buffers))
    return named


@torch.library.custom_op("hrnet_hip::shiftnet_forward_train", mutates_args=(), device_types="cuda")
def _op_shiftnet_forward_train(packed: torch.Tensor, x: torch.Tensor, params: Sequence[torch.Tensor], bn_running: Sequence[torch.Tensor],
                               momentum: float, dropout_mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
    """`shifts = regis_model(pairs)` in training (train.py:40): batch-statistics BatchNorm, the given dropout keep-mask; keeps each layer's
    pre-BatchNorm tensor and statistics in `tws` for the backward.  Functional (an op with an autograd formula must be): the updated
    running statistics come back as new tensors, in SHIFTNET_BUFFER_NAMES order, and the module copies them into its buffers."""
    new_running = [b.clone() for b in bn_running]
    theta, tws = shiftnet_forward_train(packed, _shiftnet_named(params, new_running), x, momentum=momentum, dropout_mask=dropout_mask)
    return theta, tws, new_running


@_op_shiftnet_forward_train.register_fake
def _(packed, x, params, bn_running, momentum, dropout_mask):
    nbytes = load_library().hrn_shiftnet_train_workspace_bytes(x.shape[0])
    return x.new_empty((x.shape[0], 2), dtype=torch.float32), x.new_empty((nbytes,), dtype=torch.uint8), [b.new_empty(b.shape) for b in bn_running]


@torch.library.custom_op("hrnet_hip::shiftnet_backward", mutates_args=("tws",), device_types="cuda")  # (tws also holds the backward's scratch buffers)
def _op_shiftnet_backward(params: Sequence[torch.Tensor], x: torch.Tensor,
                          dropout_mask: Optional[torch.Tensor], d_theta: torch.Tensor, tws: torch.Tensor,
                          need_input_grad: bool) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """d_theta -> (parameter gradients in SHIFTNET_PARAM_NAMES order, d_x (empty when not needed)).  The batch statistics the backward
    needs are in `tws`; the running statistics take no part (the BatchNorm weights stand in for them in the C struct)."""
    named = dict(zip(SHIFTNET_PARAM_NAMES, params))
    for k in SHIFTNET_BUFFER_NAMES:
        named[k] = named[k.rsplit(".", 1)[0] + ".weight"]
    grads = {k: torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for k, p in zip(SHIFTNET_PARAM_NAMES, params)}
    d_x = shiftnet_backward(named, grads, x, dropout_mask, d_theta.contiguous(), tws, need_input_grad=need_input_grad)
    return [grads[k] for k in SHIFTNET_PARAM_NAMES], (d_x if d_x is not None else x.new_empty((0,)))


@_op_shiftnet_backward.register_fake
def _(params, x, dropout_mask, d_theta, tws, need_input_grad):
    return [p.new_empty(p.shape, dtype=torch.float32) for p in params], (x.new_empty(x.shape) if need_input_grad else x.new_empty((0,)))


def _shiftnet_train_setup(ctx, inputs, output):
    packed, x, params, bn_running, momentum, dropout_mask = inputs
    ctx.np, ctx.has_mask = len(params), dropout_mask is not None
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(x, output[1], *params, *([dropout_mask] if dropout_mask is not None else []))


def _shiftnet_train_backward(ctx, d_theta, _d_tws, _d_running):
    x, tws, *rest = ctx.saved_tensors
    if d_theta is None:
        return None, None, [None] * ctx.np, [None] * len(SHIFTNET_BUFFER_NAMES), None, None
    params = rest[:ctx.np]
    mask = rest[ctx.np] if ctx.has_mask else None
    need_x = ctx.needs_input_grad[1]
    grads, d_x = torch.ops.hrnet_hip.shiftnet_backward(params, x, mask, d_theta, tws.data, need_x)      # (tws.data: see _hrnet_train_backward)
    return None, (d_x if need_x else None), grads, [None] * len(SHIFTNET_BUFFER_NAMES), None, None


_op_shiftnet_forward_train.register_autograd(_shiftnet_train_backward, setup_context=_shiftnet_train_setup)


@torch.library.custom_op("hrnet_hip::lanczos_shift_backward", mutates_args=(), device_types="cuda")
def _op_lanczos_shift_backward(img: torch.Tensor, shift: torch.Tensor, d_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Adjoint of lanczos_shift wrt the image and the gradient wrt the shifts (through the 7 taps per axis): (d_img, d_shift (c, 2))."""
    d_img, d_shift = lanczos_shift_backward(img, shift, d_out.contiguous(), True, True)
    full = torch.zeros_like(shift, dtype=torch.float32)
    full[:d_shift.shape[0]] = d_shift                       # `shift` may carry more rows than img has channels
    return d_img, full


@_op_lanczos_shift_backward.register_fake
def _(img, shift, d_out):
    return img.new_empty(img.shape, dtype=torch.float32), shift.new_empty(shift.shape, dtype=torch.float32)


def _lanczos_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)
    ctx.set_materialize_grads(False)


def _lanczos_backward(ctx, d_out):
    img, shift = ctx.saved_tensors
    if d_out is None:
        return None, None
    d_img, d_shift = torch.ops.hrnet_hip.lanczos_shift_backward(img, shift, d_out)
    return (d_img.to(img.dtype) if ctx.needs_input_grad[0] else None), (d_shift.to(shift.dtype) if ctx.needs_input_grad[1] else None)


_op_lanczos_shift.register_autograd(_lanczos_backward, setup_context=_lanczos_setup)


@torch.library.custom_op("hrnet_hip::get_loss_train", mutates_args=(), device_types="cuda")
def _op_get_loss_train(srs: torch.Tensor, hrs: torch.Tensor, hr_maps: torch.Tensor, metric: str, crop: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """The registered-loss tail (train.py:78-87, :183-187): (loss per sample (B,), stats (B,4) f64 = {n, bias, cMSE, 0})."""
    return get_loss_train(srs, hrs, hr_maps, metric, crop)


@_op_get_loss_train.register_fake
def _(srs, hrs, hr_maps, metric, crop):
    return srs.new_empty((srs.shape[0],), dtype=torch.float32), srs.new_empty((srs.shape[0], 4), dtype=torch.float64)


@torch.library.custom_op("hrnet_hip::get_loss_backward", mutates_args=(), device_types="cuda")
def _op_get_loss_backward(srs: torch.Tensor, hrs: torch.Tensor, hr_maps: torch.Tensor, stats: torch.Tensor, d_out: torch.Tensor,
                          metric: str, crop: int) -> torch.Tensor:
    return get_loss_backward(srs, hrs, hr_maps, stats, d_out.contiguous(), metric, crop)


@_op_get_loss_backward.register_fake
def _(srs, hrs, hr_maps, stats, d_out, metric, crop):
    return srs.new_empty(srs.shape, dtype=torch.float32)


def _loss_setup(ctx, inputs, output):
    srs, hrs, hr_maps, ctx.metric, ctx.crop = inputs
    ctx.save_for_backward(srs, hrs, hr_maps, output[1])
    ctx.set_materialize_grads(False)


def _loss_backward(ctx, d_out, _d_stats):
    srs, hrs, hr_maps, stats = ctx.saved_tensors
    if d_out is None:
        return None, None, None, None, None
    return torch.ops.hrnet_hip.get_loss_backward(srs, hrs, hr_maps, stats, d_out, ctx.metric, ctx.crop), None, None, None, None


_op_get_loss_train.register_autograd(_loss_backward, setup_context=_loss_setup)


@torch.library.custom_op("hrnet_hip::adam_step", mutates_args=("params", "exp_avg", "exp_avg_sq"), device_types="cuda")
def _op_adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr: float, beta1: float,
                  beta2: float, eps: float, weight_decay: float, step: int) -> None:
    """`optimizer.step()` of torch.optim.Adam (train.py:191) on one flat fp32 buffer, in place."""
    adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step)
