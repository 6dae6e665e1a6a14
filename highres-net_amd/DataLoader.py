"""DataLoader on the native input pipeline: same import path and names as the reference's `src/DataLoader.py`
(get_patch :16-31, ImageSet :34-49, sample_clearest :52-72, read_imageset :75-148, ImagesetDataset :153-204), so that
`from DataLoader import ImagesetDataset, ImageSet` in src/train.py:20 / src/predict.py:11 resolves here.

Directory listing, clearance loading, the clearance-softmax view sampling and the random patch corner stay in Python and
draw from numpy's global RNG in the reference's order (so a seeded run picks the same views and patch); the byte work -
PNG decode, crop, uint16 -> float32, padding - runs in libhrnet_io.so (`hrnet_hip.io_binding`).  Beyond the reference's
surface, `ImagesetDataset.load_batch()` collates a whole batch straight into (optionally pinned) buffers on a thread pool.
"""
from collections import OrderedDict
import operator
import os
import re

import numpy as np
import torch
from torch.utils.data import Dataset

from hrnet_hip import io_binding

_QM_FILE = re.compile(r"^QM(.*)\.png$", re.S)       # one quality map per LR view; the text between "QM" and ".png" is the view id


def get_patch(img, x, y, size=32):
    """Square window of the two trailing axes of `img`: rows x .. x+size-1, columns y .. y+size-1 (x is the ROW corner, as in
    the reference); leading axes (e.g. the view axis of an LR stack) are kept."""
    rows, cols = slice(x, x + size), slice(y, y + size)
    return img[..., rows, cols]


class ImageSet(OrderedDict):
    """The assets of one imageset (name, lr, hr, hr_map, clearances) as an ordered mapping whose repr lists one asset per line:
    key right-aligned to 10 columns, then shape / class / dtype for arrays and tensors, class and value for anything else."""

    def __repr__(self):
        def describe(value):
            kind = type(value).__name__
            if hasattr(value, "shape"):
                return f"{value.shape} {kind} ({value.dtype})"
            return f"{kind} ({value})"

        lines = ["name".rjust(10) + f" : {self['name']}"]
        lines.extend(key.rjust(10) + " : " + describe(value) for key, value in self.items())
        return "\n".join(lines)


def sample_clearest(clearances, n=None, beta=50, seed=None):
    """Draw `n` distinct view indices with probability softmax(beta * clearance / max clearance): beta = 0 is uniform, large beta
    approaches "the n clearest".  RNG contract (shared with the reference so that a seeded run picks the same views): an optional
    `np.random.seed(seed)`, then exactly one `np.random.choice(..., replace=False)` over the weights below, computed in the same
    floating-point order ((beta * c) / max c, exp, normalise)."""
    if seed is not None:
        np.random.seed(seed)
    weights = np.exp(np.divide(np.multiply(beta, clearances), np.max(clearances)))
    return np.random.choice(len(weights), size=n, replace=False, p=weights / np.sum(weights))


def _views_in_use_order(imset_dir, top_k, beta, seed):
    """(view ids, clearances) of an imageset in the order the model consumes them: `top_k` > 0 -> a clearance-weighted sample of
    min(top_k, L) views (sample_clearest), else all views from the clearest down."""
    ids = np.sort(np.array([m.group(1) for m in map(_QM_FILE.match, os.listdir(imset_dir)) if m]))
    score_file = os.path.join(imset_dir, "clearance.npy")
    if not os.path.isfile(score_file):
        raise Exception("please call the save_clearance.py before call DataLoader")
    scores = np.load(score_file)
    if top_k is not None and top_k > 0:
        pick = sample_clearest(scores, n=min(top_k, len(ids)), beta=beta, seed=seed)
    else:
        pick = np.flip(np.argsort(scores))
    return ids[pick], scores[pick]


def _corner(lr_side, patch_size, seed):
    """Random patch corner (row, column): an optional re-seed, then two `np.random.randint(0, side - patch)` draws, row first."""
    if seed is not None:
        np.random.seed(seed)
    limit = lr_side - patch_size
    row = np.random.randint(low=0, high=limit)
    col = np.random.randint(low=0, high=limit)
    return row, col


def read_imageset(imset_dir, create_patches=False, patch_size=64, seed=None, top_k=None, beta=0.):
    """ImageSet(name, lr uint16 (L,H,W), hr uint16 or None, hr_map bool, clearances) of one imageset directory, PNGs decoded by
    the native reader.  With `create_patches` one random `patch_size` window is cut from every LR view and the matching 3x
    window from SM / HR."""
    ids, scores = _views_in_use_order(imset_dir, top_k, beta, seed)
    asset = lambda name: os.path.join(imset_dir, name)
    lr = np.stack([io_binding.png_read(asset(f"LR{i}.png")) for i in ids]).astype(np.uint16, copy=False)
    hr_map = io_binding.png_read(asset("SM.png")) != 0
    hr = io_binding.png_read(asset("HR.png")).astype(np.uint16, copy=False) if os.path.exists(asset("HR.png")) else None
    if create_patches:
        row, col = _corner(lr.shape[1], patch_size, seed)
        lr = get_patch(lr, row, col, patch_size)
        hr_map = get_patch(hr_map, 3 * row, 3 * col, 3 * patch_size)
        if hr is not None:
            hr = get_patch(hr, 3 * row, 3 * col, 3 * patch_size)
    return ImageSet(name=os.path.basename(imset_dir), lr=np.array(lr), hr=hr, hr_map=hr_map, clearances=scores)


class ImagesetDataset(Dataset):
    """Dataset over imageset directories.  `dataset[i]` (int), `dataset["imgsetXXXX"]` (name) -> one ImageSet of float32
    tensors (lr (L,S,S), hr / hr_map (3S,3S); test imagesets keep hr = None and a bool numpy hr_map); a slice -> a list."""

    def __init__(self, imset_dir, config, seed=None, top_k=-1, beta=0.):
        super().__init__()
        self.imset_dir = imset_dir
        self.name_to_dir = dict(zip(map(os.path.basename, imset_dir), imset_dir))
        self.create_patches, self.patch_size = config["create_patches"], config["patch_size"]
        self.seed, self.top_k, self.beta = seed, top_k, beta          # seed: re-seeds numpy's global RNG per imageset when set

    def __len__(self):
        return len(self.imset_dir)

    def _resolve(self, index):
        """Directories an index stands for, and whether the caller gets a bare ImageSet (int / name) or a list (slice)."""
        if isinstance(index, str):
            return [self.name_to_dir[index]], True
        if isinstance(index, slice):
            picked = self.imset_dir[index]
            return picked, len(picked) == 1
        if isinstance(index, int):
            return [self.imset_dir[operator.index(index)]], True
        raise KeyError("index must be int, string, or slice")

    def __getitem__(self, index):
        dirs, single = self._resolve(index)
        loaded = [self._load_one(d) for d in dirs]
        return loaded[0] if single else loaded

    def _plan(self, dir_):
        """Everything random / directory-dependent for one imageset, in the reference's RNG order."""
        idx_names, clearances = _views_in_use_order(dir_, self.top_k, self.beta, self.seed)
        lr_paths = [os.path.join(dir_, f"LR{i}.png") for i in idx_names]
        lr_side = io_binding.png_info(lr_paths[0])[0]
        corner = _corner(lr_side, self.patch_size, self.seed) if self.create_patches else (0, 0)
        hr_path = os.path.join(dir_, "HR.png") if os.path.exists(os.path.join(dir_, "HR.png")) else None
        return dict(name=os.path.basename(dir_), lr_paths=lr_paths, clearances=clearances, lr_side=lr_side, corner=corner, hr=hr_path,
                    sm=os.path.join(dir_, "SM.png"))

    def _load_one(self, dir_):
        pl = self._plan(dir_)
        patch = self.patch_size if self.create_patches else 0
        out = io_binding.collate([pl["lr_paths"]], [pl["hr"]], [pl["sm"]], min_L=len(pl["lr_paths"]), lr_size=pl["lr_side"], patch=patch,
                                 corners=[pl["corner"]])
        labelled = pl["hr"] is not None
        return ImageSet(name=pl["name"], lr=torch.from_numpy(out["lrs"][0]), hr=torch.from_numpy(out["hrs"][0]) if labelled else None,
                        hr_map=torch.from_numpy(out["maps"][0]) if labelled else out["maps"][0].astype(bool), clearances=pl["clearances"])

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
