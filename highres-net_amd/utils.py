"""The helpers train.py / predict.py import from `utils` (`from utils import getImageSetDirectories, readBaselineCPSNR,
collateFunction`: reference src/train.py:22, src/predict.py:14), written from their observable behaviour (SURVEY.md 3.4,
Appendix A; reference src/utils.py:15-113 for the contract).  The plotting helpers of that module are out of scope.

`collateFunction` defines the `(lrs, alphas)` input contract of `HRNet.forward`: every imageset is cut
to its first `min_L` views or zero-padded up to `min_L`, and `alphas` flags the genuine views with 1.  Here the batch is written
view-block by view-block into ONE preallocated `(B, min_L, H, W)` buffer (optionally pinned) instead of being built from
per-sample concatenations; whole batches decoded natively take `DataLoader.ImagesetDataset.load_batch`, which fills the same
layout through `hrn_io_collate`."""
import os
from itertools import takewhile

import torch

_CHANNELS = ("RED", "NIR")          # the two spectral bands of a PROBA-V split, in the order the callers expect


def readBaselineCPSNR(path):
    """norm.csv ("<imageset> <baseline cPSNR>" per line, blank separated) -> {imageset: float}."""
    table = {}
    with open(path, "r") as fh:
        for line in fh:
            fields = line.rstrip("\r\n").split(" ")
            if len(fields) < 2 or not fields[0].strip():
                continue
            table[fields[0].strip()] = float(fields[1].strip())
    return table


def getImageSetDirectories(data_dir):
    """Every imageset directory of a split: data_dir/RED/* then data_dir/NIR/*, each band in os.listdir order."""
    return [os.path.join(data_dir, band, entry) for band in _CHANNELS for entry in os.listdir(os.path.join(data_dir, band))]


class collateFunction():
    """Callable for `DataLoader(collate_fn=...)`: list of ImageSet -> (lrs (B,min_L,H,W), alphas (B,min_L), hrs, hr_maps, names).

    hrs / hr_maps are stacked `(B, 3H, 3W)` tensors when every imageset carries its HR image; otherwise (test split) `hrs` is
    the list of HR images met before the first missing one and `hr_maps` stays the per-sample list."""

    def __init__(self, min_L=32, pin_memory=False):
        self.min_L = min_L
        self.pin_memory = pin_memory

    def __call__(self, batch):
        return self.collateFunction(batch)

    def collateFunction(self, batch):
        n_views = int(self.min_L)
        stacks = [sample["lr"] for sample in batch]
        frame = tuple(stacks[0].shape[1:])
        if any(tuple(s.shape[1:]) != frame for s in stacks):
            raise RuntimeError(f"collateFunction: the imagesets of one batch must share their LR size, got {[tuple(s.shape) for s in stacks]}")
        # a padded sample mixes its views with float32 zeros (type promotion); a batch of full stacks keeps their dtype
        dtype = stacks[0].dtype
        if any(s.shape[0] < n_views for s in stacks):
            dtype = torch.promote_types(dtype, torch.float32)
        lrs = torch.zeros((len(batch), n_views) + frame, dtype=dtype, pin_memory=self.pin_memory)
        alphas = torch.zeros((len(batch), n_views), dtype=torch.float32, pin_memory=self.pin_memory)
        for row, views in enumerate(stacks):
            real = min(views.shape[0], n_views)
            lrs[row, :real].copy_(views[:real])
            alphas[row, :real] = 1.0
        names = [sample["name"] for sample in batch]
        maps = [sample["hr_map"] for sample in batch]
        hrs = list(takewhile(lambda hr: hr is not None, (sample["hr"] for sample in batch)))
        if len(hrs) == len(batch):
            hrs, maps = torch.stack(hrs), torch.stack(maps)
        return lrs, alphas, hrs, maps, names
