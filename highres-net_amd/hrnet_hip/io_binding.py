"""ctypes binding of libhrnet_io.so (C ABI: include/hrnet_io.h): native PNG decode and batch collate for the input pipeline
(SURVEY.md section 8f row f4).  Host memory only; numpy / torch are plumbing (buffers), the byte work is in the library.
No fallback: if the library is missing these functions raise."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRNET_IO_LIB") or os.path.join(_HERE, "libhrnet_io.so")

_c = ctypes
_pp = _c.POINTER(_c.c_char_p)
_ip = _c.POINTER(_c.c_int)
SIGNATURES = {
    "hrn_io_version": (_c.c_int, []),
    "hrn_io_last_error": (_c.c_char_p, []),
    "hrn_io_png_info": (_c.c_int, [_c.c_char_p, _ip, _ip, _ip]),
    "hrn_io_png_read_u16": (_c.c_int, [_c.c_char_p, _c.c_void_p, _c.c_int, _c.c_int]),
    "hrn_io_collate": (_c.c_int, [_c.c_int, _pp, _ip, _pp, _pp, _c.c_int, _c.c_int, _c.c_int, _ip, _ip,
                                  _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int]),
}
_lib = None


class HrnetIoError(RuntimeError):
    pass


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HrnetIoError(f"{LIB_PATH} not found: build it with `python highres-net_amd/hrnet_hip/build.py` (no Python fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise HrnetIoError(f"{what} failed ({rc}): {load_library().hrn_io_last_error().decode('utf-8', 'replace')}")


def png_info(path):
    """-> (width, height, bit_depth)"""
    w, h, d = _c.c_int(), _c.c_int(), _c.c_int()
    _check(load_library().hrn_io_png_info(os.fsencode(path), _c.byref(w), _c.byref(h), _c.byref(d)), "hrn_io_png_info")
    return w.value, h.value, d.value


def png_read(path):
    """Decode a grayscale PNG -> numpy uint16 (H, W) (the stored sample values)."""
    w, h, _ = png_info(path)
    out = np.empty((h, w), np.uint16)
    _check(load_library().hrn_io_png_read_u16(os.fsencode(path), out.ctypes.data_as(_c.c_void_p), w, h), "hrn_io_png_read_u16")
    return out


def _strs(paths):
    arr = (_c.c_char_p * len(paths))()
    for i, p in enumerate(paths):
        arr[i] = None if p is None else os.fsencode(p)
    return arr


def collate(lr_paths_per_set, hr_paths, sm_paths, min_L, lr_size, patch=0, corners=None, out=None, n_threads=0):
    """lr_paths_per_set: list (one per imageset) of lists of LR files in use order; hr_paths: list of paths / None entries
    or None; sm_paths: list of paths; corners: list of (x, y) = (row, column) per imageset when patch > 0.
    out: optional dict of preallocated float32 buffers 'lrs' (B,min_L,S,S), 'alphas' (B,min_L), 'hrs', 'maps' (B,3S,3S) -
    numpy arrays or CPU torch tensors (e.g. pinned).  Returns that dict (numpy arrays when it allocates)."""
    lib = load_library()
    B = len(lr_paths_per_set)
    S = patch if patch > 0 else lr_size
    have_hr = hr_paths is not None and any(p is not None for p in hr_paths)
    if out is None:
        out = dict(lrs=np.empty((B, min_L, S, S), np.float32), alphas=np.empty((B, min_L), np.float32),
                   hrs=np.empty((B, 3 * S, 3 * S), np.float32) if have_hr else None, maps=np.empty((B, 3 * S, 3 * S), np.float32))

    def ptr(t, shape):
        if t is None:
            return None
        if tuple(t.shape) != shape:
            raise ValueError(f"buffer shape {tuple(t.shape)} != {shape}")
        if hasattr(t, "data_ptr"):                                # torch CPU tensor
            if t.is_cuda or not t.is_contiguous() or str(t.dtype) != "torch.float32":
                raise ValueError("collate buffers must be contiguous float32 host tensors")
            return _c.c_void_p(t.data_ptr())
        if t.dtype != np.float32 or not t.flags["C_CONTIGUOUS"]:
            raise ValueError("collate buffers must be C-contiguous float32 arrays")
        return t.ctypes.data_as(_c.c_void_p)

    flat = [p for views in lr_paths_per_set for p in views]
    nv = (_c.c_int * B)(*[len(v) for v in lr_paths_per_set])
    px = py = None
    if patch > 0:
        px = (_c.c_int * B)(*[int(c[0]) for c in corners])
        py = (_c.c_int * B)(*[int(c[1]) for c in corners])
    _check(lib.hrn_io_collate(B, _strs(flat), nv, _strs(hr_paths) if have_hr else None, _strs(sm_paths), int(min_L), int(lr_size),
                              int(patch), px, py, ptr(out["lrs"], (B, min_L, S, S)), ptr(out["alphas"], (B, min_L)),
                              ptr(out.get("hrs") if have_hr else None, (B, 3 * S, 3 * S)), ptr(out["maps"], (B, 3 * S, 3 * S)),
                              int(n_threads)), "hrn_io_collate")
    return out
