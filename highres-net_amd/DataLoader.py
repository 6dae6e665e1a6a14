"""DataLoader on the native input pipeline: same import path and names as the reference's `src/DataLoader.py`
(get_patch :16-31, ImageSet :34-49, sample_clearest :52-72, read_imageset :75-148, ImagesetDataset :153-204), so that
`from DataLoader import ImagesetDataset, ImageSet` in src/train.py:20 / src/predict.py:11 resolves here.

Directory listing, clearance loading, the clearance-softmax view sampling and the random patch corner stay in Python and
draw from numpy's global RNG in the reference's order (so a seeded run picks the same views and patch); the byte work -
PNG decode, crop, uint16 -> float32, padding - runs in libhrnet_io.so (`hrnet_hip.io_binding`).  Beyond the reference's
surface, `ImagesetDataset.load_batch()` collates a whole batch straight into (optionally pinned) buffers on a thread pool.
"""
from collections import OrderedDict
import glob
from os.path import basename, exists, isfile, join

import numpy as np
import torch
from torch.utils.data import Dataset

from hrnet_hip import io_binding


def get_patch(img, x, y, size=32):
    """img[..., x:x+size, y:y+size]: x is the row corner, y the column corner (the reference's naming)."""
    return img[..., x:(x + size), y:(y + size)]


class ImageSet(OrderedDict):
    """OrderedDict grouping the assets of an imageset, with the reference's pretty-print."""

    def __repr__(self):
        info = f"{'name':>10} : {self['name']}"
        for name, v in self.items():
            if hasattr(v, "shape"):
                info += f"\n{name:>10} : {v.shape} {v.__class__.__name__} ({v.dtype})"
            else:
                info += f"\n{name:>10} : {v.__class__.__name__} ({v})"
        return info


def sample_clearest(clearances, n=None, beta=50, seed=None):
    """Indices of `n` views drawn without replacement with probability softmax(beta * clearance / max clearance)."""
    if seed is not None:
        np.random.seed(seed)
    e_c = np.exp(beta * clearances / clearances.max())
    p = e_c / e_c.sum()
    return np.random.choice(range(len(p)), size=n, p=p, replace=False)


def _select(imset_dir, top_k, beta, seed):
    """View names in use order + their clearances (DataLoader.py:97-119)."""
    idx_names = np.sort(np.array([basename(path)[2:-4] for path in glob.glob(join(imset_dir, "QM*.png"))]))
    if not isfile(join(imset_dir, "clearance.npy")):
        raise Exception("please call the save_clearance.py before call DataLoader")
    clearances = np.load(join(imset_dir, "clearance.npy"))
    if top_k is not None and top_k > 0:
        top_k = min(top_k, len(idx_names))
        i_samples = sample_clearest(clearances, n=top_k, beta=beta, seed=seed)
        return idx_names[i_samples], clearances[i_samples]
    order = np.argsort(clearances)[::-1]
    return idx_names[order], clearances[order]


def _corner(lr_side, patch_size, seed):
    """Random patch corner (DataLoader.py:130-136): two randint draws after an optional re-seed."""
    if seed is not None:
        np.random.seed(seed)
    x = np.random.randint(low=0, high=lr_side - patch_size)
    y = np.random.randint(low=0, high=lr_side - patch_size)
    return x, y


def read_imageset(imset_dir, create_patches=False, patch_size=64, seed=None, top_k=None, beta=0.):
    """ImageSet(name, lr uint16 (L,H,W), hr uint16 or None, hr_map bool, clearances) - the reference's return value, decoded
    by the native PNG reader."""
    idx_names, clearances = _select(imset_dir, top_k, beta, seed)
    lr_images = np.array([io_binding.png_read(join(imset_dir, f"LR{i}.png")) for i in idx_names], dtype=np.uint16)
    hr_map = io_binding.png_read(join(imset_dir, "SM.png")).astype(bool)
    hr = io_binding.png_read(join(imset_dir, "HR.png")).astype(np.uint16) if exists(join(imset_dir, "HR.png")) else None
    if create_patches:
        x, y = _corner(lr_images[0].shape[0], patch_size, seed)
        lr_images = get_patch(lr_images, x, y, patch_size)
        hr_map = get_patch(hr_map, x * 3, y * 3, patch_size * 3)
        if hr is not None:
            hr = get_patch(hr, x * 3, y * 3, patch_size * 3)
    return ImageSet(name=basename(imset_dir), lr=np.array(lr_images), hr=hr, hr_map=hr_map, clearances=clearances)


class ImagesetDataset(Dataset):
    """Dataset over imageset directories; `__getitem__` returns the reference's ImageSet of float32 tensors."""

    def __init__(self, imset_dir, config, seed=None, top_k=-1, beta=0.):
        super().__init__()
        self.imset_dir = imset_dir
        self.name_to_dir = {basename(im_dir): im_dir for im_dir in imset_dir}
        self.create_patches = config["create_patches"]
        self.patch_size = config["patch_size"]
        self.seed = seed
        self.top_k = top_k
        self.beta = beta

    def __len__(self):
        return len(self.imset_dir)

    def _plan(self, dir_):
        """Everything random / directory-dependent for one imageset, in the reference's RNG order."""
        idx_names, clearances = _select(dir_, self.top_k, self.beta, self.seed)
        lr_paths = [join(dir_, f"LR{i}.png") for i in idx_names]
        lr_side = io_binding.png_info(lr_paths[0])[0]
        corner = _corner(lr_side, self.patch_size, self.seed) if self.create_patches else (0, 0)
        hr_path = join(dir_, "HR.png") if exists(join(dir_, "HR.png")) else None
        return dict(name=basename(dir_), lr_paths=lr_paths, clearances=clearances, lr_side=lr_side, corner=corner, hr=hr_path,
                    sm=join(dir_, "SM.png"))

    def _load_one(self, dir_):
        pl = self._plan(dir_)
        patch = self.patch_size if self.create_patches else 0
        out = io_binding.collate([pl["lr_paths"]], [pl["hr"]], [pl["sm"]], min_L=len(pl["lr_paths"]), lr_size=pl["lr_side"], patch=patch,
                                 corners=[pl["corner"]])
        imset = ImageSet(name=pl["name"], lr=torch.from_numpy(out["lrs"][0]),
                         hr=torch.from_numpy(out["hrs"][0]) if pl["hr"] is not None else None,
                         hr_map=torch.from_numpy(out["maps"][0]) if pl["hr"] is not None else out["maps"][0].astype(bool),
                         clearances=pl["clearances"])
        return imset

    def __getitem__(self, index):
        if isinstance(index, int):
            dirs = [self.imset_dir[index]]
        elif isinstance(index, str):
            dirs = [self.name_to_dir[index]]
        elif isinstance(index, slice):
            dirs = self.imset_dir[index]
        else:
            raise KeyError("index must be int, string, or slice")
        imsets = [self._load_one(d) for d in dirs]
        return imsets[0] if len(imsets) == 1 else imsets

    def load_batch(self, indices, min_L, pin_memory=False, n_threads=0):
        """One collated batch (padded_lr (B,min_L,S,S), alphas (B,min_L), hrs (B,3S,3S) or [], hr_maps (B,3S,3S), names) decoded
        straight into (optionally pinned) buffers by the native thread pool: __getitem__ + collateFunction in one call."""
        plans = [self._plan(self.imset_dir[i] if isinstance(i, int) else self.name_to_dir[i]) for i in indices]
        side = plans[0]["lr_side"]
        if any(p["lr_side"] != side for p in plans):
            raise ValueError("imagesets of one batch must share the LR size")
        patch = self.patch_size if self.create_patches else 0
        S = patch if patch else side
        B = len(plans)
        have_hr = all(p["hr"] is not None for p in plans)
        mk = lambda *shape: torch.empty(shape, dtype=torch.float32, pin_memory=pin_memory)
        out = dict(lrs=mk(B, min_L, S, S), alphas=mk(B, min_L), hrs=mk(B, 3 * S, 3 * S) if have_hr else None, maps=mk(B, 3 * S, 3 * S))
        io_binding.collate([p["lr_paths"] for p in plans], [p["hr"] for p in plans] if have_hr else None, [p["sm"] for p in plans],
                           min_L=min_L, lr_size=side, patch=patch, corners=[p["corner"] for p in plans], out=out, n_threads=n_threads)
        return out["lrs"], out["alphas"], out["hrs"] if have_hr else [], out["maps"], [p["name"] for p in plans]


class BatchPrefetcher:
    """Iterate over whole batches of an ImagesetDataset with the next batch's decode and host-to-device copy running
    while the caller computes on the current one.

        for lrs, alphas, hrs, hr_maps, names in BatchPrefetcher(dataset, batches, min_L, device="cuda"):
            ...

    `batches` is a sequence of index lists (what a BatchSampler yields).  A worker thread decodes batch n+1 with
    `load_batch(pin_memory=True)` - the native thread pool does the PNG work - and, for a CUDA device, enqueues the copies
    on a private stream; the consumer's stream waits on that copy's event only when it takes the batch, so PCIe traffic and
    decode overlap the kernels of batch n.  Tensors are handed over with `record_stream`, i.e. their memory is not reused
    before the consumer's queued work has finished.  With device=None or "cpu" it is a plain background decoder.
    Errors raised by the worker are re-raised in the consumer at the batch they belong to."""

    def __init__(self, dataset, batches, min_L, device=None, depth=2, n_threads=0):
        import queue
        import threading
        self.dataset, self.batches, self.min_L = dataset, [list(b) for b in batches], int(min_L)
        self.device = torch.device(device) if device is not None else None
        self.on_gpu = self.device is not None and self.device.type == "cuda"
        if self.on_gpu and not torch.cuda.is_available():
            raise RuntimeError("BatchPrefetcher: device is cuda but no GPU is available")
        self.n_threads = n_threads
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._work, name="hrn-batch-prefetch", daemon=True)
        self._started = False

    def __len__(self):
        return len(self.batches)

    def _work(self):
        stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        try:
            for idx in self.batches:
                if self._stop.is_set():
                    return
                try:
                    lrs, alphas, hrs, maps, names = self.dataset.load_batch(idx, self.min_L, pin_memory=self.on_gpu,
                                                                            n_threads=self.n_threads)
                    event = None
                    if self.on_gpu:
                        with torch.cuda.stream(stream):
                            lrs = lrs.to(self.device, non_blocking=True)
                            alphas = alphas.to(self.device, non_blocking=True)
                            maps = maps.to(self.device, non_blocking=True)
                            if isinstance(hrs, torch.Tensor):
                                hrs = hrs.to(self.device, non_blocking=True)
                            event = torch.cuda.Event()
                            event.record(stream)
                    item = ("ok", (lrs, alphas, hrs, maps, names), event, stream)
                except Exception as exc:                     # handed to the consumer, in order
                    item = ("err", exc, None, None)
                while not self._stop.is_set():
                    try:
                        self._q.put(item, timeout=0.1)
                        break
                    except Exception:
                        continue
                if item[0] == "err":
                    return
        finally:
            while not self._stop.is_set():
                try:
                    self._q.put(("end", None, None, None), timeout=0.1)
                    break
                except Exception:
                    continue

    def __iter__(self):
        if self._started:
            raise RuntimeError("BatchPrefetcher can be iterated once")
        self._started = True
        self._thread.start()
        try:
            while True:
                kind, payload, event, stream = self._q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise payload
                if event is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(event)
                    for t in payload[:4]:
                        if isinstance(t, torch.Tensor):
                            t.record_stream(cur)
                yield payload
        finally:
            self.close()

    def close(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=5.0)
