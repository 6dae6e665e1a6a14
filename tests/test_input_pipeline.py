"""CPU: the native input pipeline (include/hrnet_io.h, highres-net_amd/DataLoader.py, utils.py).

There is no PROBA-V data in the container, so the fixtures are synthetic imagesets of the dataset's layout (LRxxx.png /
QMxxx.png per view, SM.png, HR.png, clearance.npy) written here with Pillow.  The checks are against Pillow's own decode
and a numpy / torch restatement of the reference's loader logic (DataLoader.py:72-148,195-199; utils.py:85-95) - bit-exact,
this is integer / byte work.  The reference's loader itself cannot be imported (skimage is not installed): parity is pinned
to these restatements, not to a reference run.
"""
import os
import zlib

import numpy as np
import pytest
import torch

PIL_Image = pytest.importorskip("PIL.Image")

import DataLoader as DL
import utils as U
from hrnet_hip import io_binding


def _write_imageset(root, name, n_views, lr=128, with_hr=True, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    d = os.path.join(root, name)
    os.makedirs(d)
    for v in range(n_views):
        a = rng.integers(0, 16000, (lr, lr), dtype=np.uint16)
        a[3:9, 5:40] = 65535                                   # saturated patch: exercises the full 16-bit range
        PIL_Image.fromarray(a).save(os.path.join(d, f"LR{v:03d}.png"), compress_level=int(rng.integers(1, 9)))
        PIL_Image.fromarray((rng.random((lr, lr)) > 0.2).astype(np.uint8) * 255).save(os.path.join(d, f"QM{v:03d}.png"))
    sm = (rng.random((3 * lr, 3 * lr)) > 0.1).astype(np.uint8) * 255
    PIL_Image.fromarray(sm).save(os.path.join(d, "SM.png"))
    if with_hr:
        PIL_Image.fromarray(rng.integers(0, 20000, (3 * lr, 3 * lr), dtype=np.uint16)).save(os.path.join(d, "HR.png"))
    np.save(os.path.join(d, "clearance.npy"), rng.random(n_views))
    return d


def _pil(path):
    return np.array(PIL_Image.open(path))


def test_png_decoder_is_bit_exact(tmp_path):
    rng = np.random.Generator(np.random.PCG64(1))
    cases = {"u16": rng.integers(0, 65536, (37, 53), dtype=np.uint16), "u8": rng.integers(0, 256, (40, 33), dtype=np.uint8),
             "smooth16": (np.add.outer(np.arange(64), np.arange(80)) * 257 % 65536).astype(np.uint16)}
    for name, a in cases.items():
        for level in (0, 1, 6, 9):
            p = str(tmp_path / f"{name}_{level}.png")
            PIL_Image.fromarray(a).save(p, compress_level=level)
            assert io_binding.png_info(p) == (a.shape[1], a.shape[0], a.dtype.itemsize * 8)
            assert np.array_equal(io_binding.png_read(p), a.astype(np.uint16)), (name, level)
    # 1-bit mask (Pillow mode '1') and every PNG filter type written by hand
    m = rng.random((19, 21)) > 0.5
    p = str(tmp_path / "bit.png")
    PIL_Image.fromarray(m).save(p)
    assert io_binding.png_info(p)[2] == 1 and np.array_equal(io_binding.png_read(p), m.astype(np.uint16))
    a = cases["u16"]
    for ft in range(5):
        rows = []
        prev = np.zeros(a.shape[1] * 2, np.int32)
        for y in range(a.shape[0]):
            cur = np.frombuffer(a[y].astype(">u2").tobytes(), np.uint8).astype(np.int32)
            left = np.concatenate([[0, 0], cur[:-2]])
            ul = np.concatenate([[0, 0], prev[:-2]])
            if ft == 0: f = cur
            elif ft == 1: f = cur - left
            elif ft == 2: f = cur - prev
            elif ft == 3: f = cur - ((left + prev) >> 1)
            else:
                pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
                f = cur - pred
            rows.append(bytes([ft]) + (f & 255).astype(np.uint8).tobytes())
            prev = cur
        def chunk(tag, data):
            import struct
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
        import struct
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", a.shape[1], a.shape[0], 16, 0, 0, 0, 0))
        comp = zlib.compress(b"".join(rows))
        png += chunk(b"IDAT", comp[:100]) + chunk(b"IDAT", comp[100:]) + chunk(b"IEND", b"")      # split IDAT
        p = str(tmp_path / f"filter{ft}.png")
        open(p, "wb").write(png)
        assert np.array_equal(_pil(p), a)
        assert np.array_equal(io_binding.png_read(p), a), ft
    with pytest.raises(io_binding.HrnetIoError):
        io_binding.png_read(str(tmp_path / "missing.png"))
    PIL_Image.fromarray(rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)).save(str(tmp_path / "rgb.png"))
    with pytest.raises(io_binding.HrnetIoError, match="grayscale"):
        io_binding.png_read(str(tmp_path / "rgb.png"))


def _restated_read(d, create_patches, patch_size, seed, top_k, beta):
    """The reference's read_imageset + __getitem__ conversion, restated on Pillow / numpy."""
    names = np.sort(np.array([os.path.basename(p)[2:-4] for p in __import__("glob").glob(os.path.join(d, "QM*.png"))]))
    cl = np.load(os.path.join(d, "clearance.npy"))
    if top_k is not None and top_k > 0:
        k = min(top_k, len(names))
        if seed is not None:
            np.random.seed(seed)
        e = np.exp(beta * cl / cl.max())
        i = np.random.choice(range(len(e)), size=k, p=e / e.sum(), replace=False)
        names, cl = names[i], cl[i]
    else:
        o = np.argsort(cl)[::-1]
        names, cl = names[o], cl[o]
    lr = np.array([_pil(os.path.join(d, f"LR{i}.png")) for i in names], dtype=np.uint16)
    sm = _pil(os.path.join(d, "SM.png")).astype(bool)
    hr = _pil(os.path.join(d, "HR.png")).astype(np.uint16) if os.path.exists(os.path.join(d, "HR.png")) else None
    if create_patches:
        if seed is not None:
            np.random.seed(seed)
        x = np.random.randint(low=0, high=lr[0].shape[0] - patch_size)
        y = np.random.randint(low=0, high=lr[0].shape[1] - patch_size)
        lr = lr[..., x:x + patch_size, y:y + patch_size]
        sm = sm[3 * x:3 * x + 3 * patch_size, 3 * y:3 * y + 3 * patch_size]
        if hr is not None:
            hr = hr[3 * x:3 * x + 3 * patch_size, 3 * y:3 * y + 3 * patch_size]
    f = lambda a: (a.astype(np.float64) / 65535.0).astype(np.float32)        # skimage.img_as_float(uint16).astype(float32)
    return dict(lr_u16=lr, lr=f(lr), hr=None if hr is None else f(hr), sm=sm, cl=cl)


@pytest.mark.parametrize("create_patches,top_k,beta,seed", [(False, -1, 0.0, None), (True, 5, 50.0, 7), (True, -1, 0.0, 3)])
def test_dataset_matches_restated_loader(tmp_path, create_patches, top_k, beta, seed):
    d = _write_imageset(str(tmp_path), "imgset0001", 9, seed=4)
    cfg = {"create_patches": create_patches, "patch_size": 64}
    want = _restated_read(d, create_patches, 64, seed, top_k, beta)
    ims = DL.read_imageset(d, create_patches=create_patches, patch_size=64, seed=seed, top_k=top_k, beta=beta)
    assert ims["name"] == "imgset0001" and ims["lr"].dtype == np.uint16 and ims["hr_map"].dtype == bool
    assert np.array_equal(ims["lr"], want["lr_u16"]) and np.array_equal(ims["hr_map"], want["sm"]) and np.array_equal(ims["clearances"], want["cl"])
    ds = DL.ImagesetDataset([d], cfg, seed=seed, top_k=top_k, beta=beta)
    item = ds[0]
    assert isinstance(item, DL.ImageSet) and item["lr"].dtype == torch.float32
    assert np.array_equal(item["lr"].numpy(), want["lr"])                     # bit-exact float32
    assert np.array_equal(item["hr"].numpy(), want["hr"])
    assert np.array_equal(item["hr_map"].numpy(), want["sm"].astype(np.float32))
    assert ds["imgset0001"]["name"] == "imgset0001" and "lr" in repr(item)


def test_collate_and_native_batch(tmp_path):
    dirs = [_write_imageset(str(tmp_path), f"imgset{i:04d}", n, seed=10 + i) for i, n in enumerate((3, 12, 7))]
    cfg = {"create_patches": True, "patch_size": 64}
    ds = DL.ImagesetDataset(dirs, cfg, seed=5, top_k=-1)
    items = [ds[i] for i in range(3)]
    lrs, alphas, hrs, maps, names = U.collateFunction(min_L=8)(items)
    assert lrs.shape == (3, 8, 64, 64) and hrs.shape == (3, 192, 192) and names == ["imgset0000", "imgset0001", "imgset0002"]
    assert alphas.tolist() == [[1] * 3 + [0] * 5, [1] * 8, [1] * 7 + [0]]
    assert float(lrs[0, 3:].abs().max()) == 0.0 and torch.equal(lrs[1], items[1]["lr"][:8])
    # the fused native path writes the same batch straight into its buffers
    b_lrs, b_alphas, b_hrs, b_maps, b_names = ds.load_batch([0, 1, 2], min_L=8, n_threads=3)
    assert torch.equal(b_lrs, lrs) and torch.equal(b_alphas, alphas) and torch.equal(b_hrs, hrs) and torch.equal(b_maps, maps)
    assert b_names == names
    # test split: no HR -> hr batch stays a list, maps stay per-sample arrays (utils.py:97-111)
    t = _write_imageset(str(tmp_path), "imgset9000", 4, with_hr=False, seed=99)
    ds2 = DL.ImagesetDataset([t], {"create_patches": False, "patch_size": 64}, top_k=-1)
    lrs2, alphas2, hrs2, maps2, _ = U.collateFunction(min_L=6)([ds2[0]])
    assert lrs2.shape == (1, 6, 128, 128) and hrs2 == [] and isinstance(maps2, list) and maps2[0].dtype == bool
    b = ds2.load_batch([0], min_L=6)
    assert torch.equal(b[0], lrs2) and b[2] == [] and np.array_equal(b[3][0].numpy() > 0, maps2[0])
    # errors are loud
    with pytest.raises(io_binding.HrnetIoError, match="expected"):
        io_binding.collate([[os.path.join(dirs[0], "LR000.png")]], None, [os.path.join(dirs[0], "SM.png")], min_L=1, lr_size=64)


def test_utils_helpers(tmp_path):
    for ch in ("RED", "NIR"):
        os.makedirs(tmp_path / "train" / ch / f"imgset_{ch}")
    got = U.getImageSetDirectories(str(tmp_path / "train"))
    assert [os.path.basename(g) for g in got] == ["imgset_RED", "imgset_NIR"]
    (tmp_path / "norm.csv").write_text("imgset0000 52.5\nimgset0001 48.25\n")
    assert U.readBaselineCPSNR(str(tmp_path / "norm.csv")) == {"imgset0000": 52.5, "imgset0001": 48.25}


def test_io_abi_matches_header():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "hrnet_io.h")).read()
    declared = set(re.findall(r"\b(hrn_io_\w+)\s*\(", hdr))
    assert declared == set(io_binding.SIGNATURES), declared ^ set(io_binding.SIGNATURES)
    lib = io_binding.load_library()
    for name in declared:
        assert hasattr(lib, name)


def test_batch_prefetcher_cpu(tmp_path):
    """Background decode of whole batches: same tensors as load_batch, in order; worker errors surface in the consumer."""
    dirs = [_write_imageset(str(tmp_path), f"imgset{i:04d}", n, seed=20 + i) for i, n in enumerate((4, 9, 6, 5))]
    ds = DL.ImagesetDataset(dirs, {"create_patches": True, "patch_size": 64}, seed=11, top_k=-1)
    batches = [[0, 1], [2, 3], [1]]
    want = [ds.load_batch(b, min_L=8) for b in batches]
    got = list(DL.BatchPrefetcher(ds, batches, min_L=8, device="cpu", depth=1))
    assert len(got) == 3
    for g, w in zip(got, want):
        assert all(torch.equal(a, b) for a, b in zip(g[:4], w[:4])) and g[4] == w[4]
    # an early exit stops the worker; a second iteration is refused
    pf = DL.BatchPrefetcher(ds, batches * 4, min_L=8)
    it = iter(pf)
    next(it)
    it.close()
    assert not pf._thread.is_alive()
    with pytest.raises(RuntimeError, match="once"):
        iter(pf).__next__()
    # a broken imageset raises at its batch, after the good ones
    os.remove(os.path.join(dirs[3], "SM.png"))
    ds_bad = DL.ImagesetDataset(dirs, {"create_patches": True, "patch_size": 64}, seed=11, top_k=-1)
    seen = 0
    with pytest.raises(Exception):
        for _ in DL.BatchPrefetcher(ds_bad, [[0, 1], [2, 3]], min_L=8):
            seen += 1
    assert seen == 1
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU"):
            DL.BatchPrefetcher(ds, batches, min_L=8, device="cuda")


@pytest.mark.gpu
def test_batch_prefetcher_gpu(tmp_path):
    """Copies on the prefetcher's stream are ordered before the consumer's kernels: device batches equal the host ones."""
    dirs = [_write_imageset(str(tmp_path), f"imgset{i:04d}", n, seed=30 + i) for i, n in enumerate((4, 9, 6, 5))]
    ds = DL.ImagesetDataset(dirs, {"create_patches": True, "patch_size": 64}, seed=2, top_k=-1)
    batches = [[0, 1], [2, 3], [3, 0], [1, 2]] * 3
    want = [ds.load_batch(b, min_L=8) for b in batches]
    n = 0
    for g, w in zip(DL.BatchPrefetcher(ds, batches, min_L=8, device="cuda"), want):
        assert g[0].is_cuda and g[1].is_cuda and g[3].is_cuda
        s = (g[0].sum() + g[2].sum()).item()                 # consumer-stream work on the fresh tensors
        assert abs(s - (w[0].sum() + w[2].sum()).item()) <= 1e-3 * abs(s) + 1e-3
        assert torch.equal(g[0].cpu(), w[0]) and torch.equal(g[1].cpu(), w[1]) and torch.equal(g[2].cpu(), w[2]) and torch.equal(g[3].cpu(), w[3])
        n += 1
    assert n == len(batches)
