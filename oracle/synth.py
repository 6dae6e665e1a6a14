"""Synthetic PROBA-V-like scenes.  numpy PCG64 only, so inputs
are reproducible on the GPU box.  Neither the PROBA-V data nor any checkpoint ships
with the reference, so every parity / bench input is synthetic (stated in bench.py's
`data` field).

The produced `(lrs, alphas)` follow the contract of the reference's collate function
(/root/reference/src/utils.py:63-113): views beyond `n_real` are all-zero and carry
alpha = 0; real views carry alpha = 1.
"""
import numpy as np


def _smooth_field(rng, n, n_waves=8):
    y, x = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    f = np.zeros((n, n))
    for _ in range(n_waves):
        fx, fy = rng.uniform(-3, 3, 2) * (2 * np.pi / n)
        ph = rng.uniform(0, 2 * np.pi)
        f += rng.uniform(0.3, 1.0) * np.sin(fx * x + fy * y + ph)
    # 1/f-ish texture: a few octaves of box-filtered white noise
    for octave in (2, 4, 8):
        m = max(n // octave, 1)
        coarse = rng.standard_normal((m + 1, m + 1))
        f += 0.5 / octave * np.kron(coarse, np.ones((octave, octave)))[:n, :n]
    f -= f.min()
    f /= max(f.max(), 1e-12)
    return f


def make_scene(rng, lr_size, n_views, n_real=None, hr_range=0.25):
    """One scene: hr (3S,3S), lrs (V,S,S), alphas (V,).  float32."""
    s = lr_size
    n_real = n_views if n_real is None else n_real
    hr = _smooth_field(rng, 3 * s + 6) * hr_range          # 3 px margin for sub-pixel shifts
    lrs = np.zeros((n_views, s, s), dtype=np.float32)
    alphas = np.zeros((n_views,), dtype=np.float32)
    for v in range(n_real):
        dy, dx = rng.uniform(-1, 1, 2)
        iy, ix = int(np.floor(dy)), int(np.floor(dx))
        fy, fx = dy - iy, dx - ix
        # bilinear sub-pixel shift of the HR field, then 3x3 box down-sampling
        a = hr[3 + iy:3 + iy + 3 * s + 1, 3 + ix:3 + ix + 3 * s + 1]
        sh = ((1 - fy) * (1 - fx) * a[:-1, :-1] + (1 - fy) * fx * a[:-1, 1:]
              + fy * (1 - fx) * a[1:, :-1] + fy * fx * a[1:, 1:])
        lr = sh.reshape(s, 3, s, 3).mean(axis=(1, 3))
        lr = lr + 0.002 * rng.standard_normal(lr.shape)
        lr = np.clip(np.round(lr * 65535.0), 0, 65535) / 65535.0   # uint16 quantisation (DataLoader.py:195)
        lrs[v] = lr.astype(np.float32)
        alphas[v] = 1.0
    return hr[3:3 + 3 * s, 3:3 + 3 * s].astype(np.float32), lrs, alphas


def make_batch(seed, batch, n_views, lr_size, n_real=None):
    """lrs (B,V,S,S) f32, alphas (B,V) f32, hrs (B,3S,3S) f32.

    `n_real` may be an int (same for every sample), a list of length B, or None (= V).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    lrs = np.zeros((batch, n_views, lr_size, lr_size), np.float32)
    alphas = np.zeros((batch, n_views), np.float32)
    hrs = np.zeros((batch, 3 * lr_size, 3 * lr_size), np.float32)
    for b in range(batch):
        nr = n_real[b] if isinstance(n_real, (list, tuple)) else n_real
        hrs[b], lrs[b], alphas[b] = make_scene(rng, lr_size, n_views, nr)
    return lrs, alphas, hrs


def fast_batch(seed, batch, n_views, lr_size):
    """Cheap full-size inputs for throughput runs (same value range; no scene model)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = rng.random((batch, 1, lr_size, lr_size), dtype=np.float32) * 0.25
    jit = rng.random((batch, n_views, lr_size, lr_size), dtype=np.float32) * 0.01
    lrs = np.round((base + jit) * 65535.0).astype(np.float32) / np.float32(65535.0)
    return lrs.astype(np.float32), np.ones((batch, n_views), np.float32)
