"""numpy restatement of the HighRes-net hot path.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  Written from the algorithm description in SURVEY.md appendix A
and the reference sources cited per function; contains no reference code.

All functions take/return NCHW numpy arrays like the reference modules and compute in
`dtype` (float64 by default: the oracle is then *more* exact than the fp32 reference, so
both the reference goldens and the HIP path are compared against the same fp64 truth).
"""
import numpy as np


# ----------------------------------------------------------------------------- primitives
def prelu(x, a):
    """nn.PReLU with a single shared slope: max(0,x) + a*min(0,x) (HRNet.py:19,21,53,97,151)."""
    a = np.asarray(a).reshape(()).astype(x.dtype)
    return np.where(x >= 0, x, a * x)


def conv3x3(x, w, b=None):
    """nn.Conv2d(k=3, padding=1): cross-correlation, zero pad 1, stride 1 (HRNet.py:18,20,52,59,95).

    x (N,Ci,H,W), w (Co,Ci,3,3) OIHW, b (Co,) -> (N,Co,H,W).  One GEMM per tap.
    """
    n, ci, h, wd = x.shape
    co = w.shape[0]
    xp = np.zeros((n, ci, h + 2, wd + 2), dtype=x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((n, co, h * wd), dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            xs = np.ascontiguousarray(xp[:, :, ky:ky + h, kx:kx + wd]).reshape(n, ci, h * wd)
            out += np.matmul(w[:, :, ky, kx].astype(x.dtype), xs)
    if b is not None:
        out += b.astype(x.dtype)[None, :, None]
    return out.reshape(n, co, h, wd)


def residual_block(x, st, prefix):
    """x + PReLU(conv(PReLU(conv(x)))): second PReLU sits inside the branch (HRNet.py:17-22,32-33)."""
    t = prelu(conv3x3(x, st[prefix + ".block.0.weight"], st[prefix + ".block.0.bias"]), st[prefix + ".block.1.weight"])
    t = prelu(conv3x3(t, st[prefix + ".block.2.weight"], st[prefix + ".block.2.bias"]), st[prefix + ".block.3.weight"])
    return x + t


# ----------------------------------------------------------------------------- HRNet stages
def reference_frame(lrs):
    """Per-pixel LOWER median over the first min(V,9) views, zero-padded views included
    (HRNet.py:200; torch.median returns the lower middle for even counts)."""
    n = min(lrs.shape[1], 9)
    srt = np.sort(lrs[:, :n], axis=1)
    return srt[:, (n - 1) // 2]


def stack_input(lrs):
    """(B,V,H,W) -> (B*V, 2, H, W): channel 0 = the view, channel 1 = shared median (HRNet.py:200-204)."""
    b, v, h, w = lrs.shape
    assert h == w, "reference reinterprets (H,W) as (W,H) in its view(); square inputs only (HRNet.py:204)"
    ref = reference_frame(lrs)
    x = np.stack([lrs, np.broadcast_to(ref[:, None], lrs.shape)], axis=2)
    return x.reshape(b * v, 2, h, w)


def encoder(x, st, num_layers=2):
    """Encoder.forward (HRNet.py:62-74)."""
    x = prelu(conv3x3(x, st["encode.init_layer.0.weight"], st["encode.init_layer.0.bias"]), st["encode.init_layer.1.weight"])
    for i in range(num_layers):
        x = residual_block(x, st, f"encode.res_layers.{i}")
    return conv3x3(x, st["encode.final.0.weight"], st["encode.final.0.bias"])


def fuse_level(x, alphas, st, alpha_residual=True):
    """One halving step of RecuversiveNet.forward (HRNet.py:113-132).

    x (B,n,C,H,W), alphas (B,n) -> x' (B,n//2,C,H,W), alphas' (B,n//2).
    Pair i <-> n-parity-1-i; the odd leftover view is dropped.
    """
    b, n, c, h, w = x.shape
    parity, half = n % 2, n // 2
    alice = x[:, :half]
    bob = x[:, half:n - parity][:, ::-1]
    z = np.concatenate([alice, bob], axis=2).reshape(b * half, 2 * c, h, w)
    u = residual_block(z, st, "fuse.fuse.0")
    f = prelu(conv3x3(u, st["fuse.fuse.1.weight"], st["fuse.fuse.1.bias"]), st["fuse.fuse.2.weight"])
    f = f.reshape(b, half, c, h, w)
    if alpha_residual:
        a_bob = alphas[:, half:n - parity][:, ::-1]
        f = alice + a_bob[:, :, None, None, None].astype(x.dtype) * f
        alphas = alphas[:, :half]
    return f, alphas


def fuse(x, alphas, st, alpha_residual=True, levels_out=None):
    """RecuversiveNet.forward (HRNet.py:99-134): halve until one view is left, then mean over views."""
    while x.shape[1] // 2 > 0:
        x, alphas = fuse_level(x, alphas, st, alpha_residual)
        if levels_out is not None:
            levels_out.append(x.copy())
    return x.mean(axis=1)


def decoder(x, st):
    """Decoder.forward (HRNet.py:158-169): ConvTranspose2d(64,64,k3,s3)+PReLU, then conv1x1 64->1.

    stride == kernel => no overlap: out[b,co,3y+ky,3x+kx] = bias[co] + sum_ci x[b,ci,y,x] * W[ci,co,ky,kx].
    """
    wd = st["decode.deconv.0.weight"].astype(x.dtype)          # (Cin, Cout, 3, 3)
    b, ci, h, w = x.shape
    co = wd.shape[1]
    up = np.einsum("bihw,iokl->bohkwl", x, wd, optimize=True).reshape(b, co, 3 * h, 3 * w)
    up = up + st["decode.deconv.0.bias"].astype(x.dtype)[None, :, None, None]
    up = prelu(up, st["decode.deconv.1.weight"])
    wf = st["decode.final.weight"].astype(x.dtype).reshape(-1, co)  # (1, 64)
    out = np.einsum("oc,bchw->bohw", wf, up, optimize=True)
    return out + st["decode.final.bias"].astype(x.dtype)[None, :, None, None]


def hrnet_forward(lrs, alphas, st, alpha_residual=True, num_layers=2, dtype=np.float64, stages=None):
    """HRNet.forward (HRNet.py:186-211).  lrs (B,V,H,W), alphas (B,V) -> (B,1,3H,3W).

    If `stages` is a dict it receives the per-stage tensors (reference frame, encoder output,
    every fusion level, fused state) for staged parity tests.
    """
    lrs = np.asarray(lrs, dtype=dtype)
    alphas = np.asarray(alphas, dtype=dtype)
    b, v, h, w = lrs.shape
    x = stack_input(lrs)
    emb = encoder(x, st, num_layers).reshape(b, v, -1, h, w)
    levels = [] if stages is not None else None
    fused = fuse(emb, alphas, st, alpha_residual, levels)
    sr = decoder(fused, st)
    if stages is not None:
        stages["ref"] = reference_frame(lrs)
        stages["emb"] = emb
        stages["levels"] = levels
        stages["fused"] = fused
    return sr


# ----------------------------------------------------------------------------- ShiftNet
def batchnorm(x, st, prefix, train, eps=1e-5):
    """nn.BatchNorm2d defaults (ShiftNet.py:17 etc.): eval -> running stats; train -> biased batch stats."""
    g = st[prefix + ".weight"].astype(x.dtype)[None, :, None, None]
    be = st[prefix + ".bias"].astype(x.dtype)[None, :, None, None]
    if train:
        mu = x.mean(axis=(0, 2, 3), keepdims=True)
        var = x.var(axis=(0, 2, 3), keepdims=True)
    else:
        mu = st[prefix + ".running_mean"].astype(x.dtype)[None, :, None, None]
        var = st[prefix + ".running_var"].astype(x.dtype)[None, :, None, None]
    return (x - mu) / np.sqrt(var + eps) * g + be


def maxpool2(x):
    n, c, h, w = x.shape
    return x.reshape(n, c, h // 2, 2, w // 2, 2).max(axis=(3, 5))


def shiftnet_forward(x, st, train_bn=False, dropout_mask=None, dtype=np.float64, layers_out=None):
    """ShiftNet.forward (ShiftNet.py:49-75).  x (B,2,128,128) -> theta (B,2).

    `dropout_mask` (B,32768) of {0,1}: when given, activations are multiplied by mask/(1-p), p=0.5
    (train-mode nn.Dropout, ShiftNet.py:43,70); None = eval mode (identity).
    """
    x = np.asarray(x, dtype=dtype)
    x = x - x.mean(axis=(2, 3), keepdims=True)                       # ShiftNet.py:58
    for i in range(1, 9):
        x = conv3x3(x, st[f"layer{i}.0.weight"], st[f"layer{i}.0.bias"])
        x = np.maximum(batchnorm(x, st, f"layer{i}.1", train_bn), 0)
        if i in (2, 4, 6):
            x = maxpool2(x)
        if layers_out is not None:
            layers_out.append(x.copy())
    x = x.reshape(x.shape[0], -1)                                    # NCHW flatten, C*H*W = 32768 (ShiftNet.py:69)
    if dropout_mask is not None:
        x = x * (np.asarray(dropout_mask, dtype=dtype) * 2.0)
    x = np.maximum(x @ st["fc1.weight"].astype(dtype).T + st["fc1.bias"].astype(dtype), 0)
    return x @ st["fc2.weight"].astype(dtype).T


# ----------------------------------------------------------------------------- Lanczos
def lanczos_kernel(dx, a=3, n=7, dtype=np.float32):
    """lanczos_kernel (lanczos.py:5-43).  dx (M,1) -> taps (M,n).

    x_j = (j - (n-1)/2) - dx;  t = pi*x_j;  t == 0 -> 1e-6;  k = sin(t)/t * sin(t/a)/(t/a);  k /= sum(k).
    No |x| < a window cut (taps beyond the support keep their small values) - as the reference.
    Default dtype float32 mirrors the reference's arithmetic type (it inherits img.dtype, lanczos.py:80).
    """
    dx = np.asarray(dx, dtype=dtype).reshape(-1, 1)
    lobes = (n - 1) // 2
    x = np.linspace(-lobes, lobes, n, dtype=dtype).reshape(1, -1) - dx
    t = dtype(np.pi) * x
    t = np.where(t == 0, dtype(1e-6), t)
    k = np.sin(t) / t * (np.sin(t / dtype(a)) / (t / dtype(a)))
    return (k / k.sum(axis=1, keepdims=True)).astype(dtype)


def lanczos_shift(img, shift, p=3, a=3, n=7, dtype=np.float32):
    """lanczos_shift (lanczos.py:47-107).  img (b,c,H,W), shift (c,2) = (dy,dx) per channel.

    Per channel: ReflectionPad2d(p) -> 7x1 correlation (zero pad 3) -> 1x7 correlation (zero pad 3) -> crop p.
    For p >= 3 the zero padding only touches cropped rows/cols, so this equals a reflect-pad-3 followed
    by a valid separable correlation: out(y,x) = sum_m sum_n ky[m] kx[n] I(y+m-3, x+n-3).
    """
    assert p >= n // 2
    img = np.asarray(img, dtype=dtype)
    shift = np.asarray(shift, dtype=dtype)
    b, c, h, w = img.shape
    r = n // 2
    out = np.empty_like(img)
    for ch in range(c):
        ky = lanczos_kernel(shift[ch, 0:1], a, n, dtype)[0]
        kx = lanczos_kernel(shift[ch, 1:2], a, n, dtype)[0]
        pad = np.pad(img[:, ch], ((0, 0), (r, r), (r, r)), mode="reflect")
        tmp = np.zeros((b, h, w + 2 * r), dtype=dtype)
        for m in range(n):                                    # vertical pass first (lanczos.py:90)
            tmp += ky[m] * pad[:, m:m + h, :]
        acc = np.zeros((b, h, w), dtype=dtype)
        for m in range(n):                                    # then horizontal (lanczos.py:94)
            acc += kx[m] * tmp[:, :, m:m + w]
        out[:, ch] = acc
    return out


def shiftnet_transform(theta, images, dtype=np.float32):
    """ShiftNet.transform (ShiftNet.py:77-90): theta (B,2)=(dx,dy), images (B,1,H,W) -> (1,1,B,H,W)."""
    theta = np.asarray(theta)
    img = np.transpose(np.asarray(images), (1, 0, 2, 3))          # I.transpose(0,1) -> (1,B,H,W)
    return lanczos_shift(img, theta[:, ::-1], p=5, a=3, n=7, dtype=dtype)[:, None]


# ----------------------------------------------------------------------------- callers: loss / metric
def get_loss(srs, hrs, hr_maps, metric="cMSE"):
    """get_loss (train.py:66-87): masked, brightness-corrected MSE per sample; 'cPSNR' -> -10 log10."""
    srs, hrs, hr_maps = (np.asarray(t, dtype=np.float64) for t in (srs, hrs, hr_maps))
    if metric == "masked_MSE":
        return ((hr_maps * srs - hr_maps * hrs) ** 2).mean(axis=(1, 2))
    nclear = hr_maps.sum(axis=(1, 2))
    bright = (hr_maps * (hrs - srs)).sum(axis=(1, 2)) / nclear
    loss = (hr_maps * (srs + bright[:, None, None] - hrs) ** 2).sum(axis=(1, 2)) / nclear
    return loss if metric == "cMSE" else -10.0 * np.log10(loss)


def cpsnr(sr, hr, hr_map):
    """Evaluator.cPSNR (Evaluator.py:11-43) for float images in [0,1]; 2-D or (n,H,W)."""
    sr, hr, hr_map = (np.asarray(t, dtype=np.float64) for t in (sr, hr, hr_map))
    single = sr.ndim == 2
    if single:
        sr, hr, hr_map = sr[None], hr[None], hr_map[None]
    n_clear = hr_map.sum(axis=(1, 2))
    diff = hr - sr
    bias = (diff * hr_map).sum(axis=(1, 2)) / n_clear
    cmse = (((diff - bias[:, None, None]) * hr_map) ** 2).sum(axis=(1, 2)) / n_clear
    out = -10.0 * np.log10(cmse)
    return out[0] if single else out


def shift_cpsnr(sr, hr, hr_map, border_w=3):
    """Evaluator.shift_cPSNR (Evaluator.py:52-73): max cPSNR over the (2w+1)^2 integer offsets of hr."""
    size = sr.shape[-1] - 2 * border_w
    src = sr[..., border_w:border_w + size, border_w:border_w + size]
    best = None
    for x in range(2 * border_w + 1):
        for y in range(2 * border_w + 1):
            # get_patch(img, x, y, size) == img[..., x:x+size, y:y+size] (DataLoader.py:16-30)
            val = cpsnr(src, hr[..., x:x + size, y:y + size], hr_map[..., x:x + size, y:y + size])
            best = val if best is None else np.maximum(best, val)
    return best
