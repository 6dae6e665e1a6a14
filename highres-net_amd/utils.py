"""The three helpers of the reference's `src/utils.py` that train.py / predict.py import (`from utils import
getImageSetDirectories, readBaselineCPSNR, collateFunction`, train.py:22, predict.py:14): readBaselineCPSNR :15-29,
getImageSetDirectories :32-47, collateFunction :51-113.  The plotting helpers of that file are out of scope.

`collateFunction` keeps the reference's per-sample semantics (truncate to min_L or zero-pad, alphas 1 / 0, HR batch only when
every sample has one); for whole batches decoded natively see `DataLoader.ImagesetDataset.load_batch`."""
import csv
import os

import torch


def readBaselineCPSNR(path):
    """{'imgsetXXXX': baseline cPSNR} from the space-separated norm.csv."""
    scores = dict()
    with open(path, "r") as file:
        for row in csv.reader(file, delimiter=" "):
            scores[row[0].strip()] = float(row[1].strip())
    return scores


def getImageSetDirectories(data_dir):
    """Imageset directories under data_dir/RED and data_dir/NIR, in os.listdir order."""
    imageset_dirs = []
    for channel_dir in ["RED", "NIR"]:
        path = os.path.join(data_dir, channel_dir)
        for imageset_name in os.listdir(path):
            imageset_dirs.append(os.path.join(path, imageset_name))
    return imageset_dirs


class collateFunction():
    """Pads / truncates the low-res views of each imageset to min_L and stacks the batch."""

    def __init__(self, min_L=32):
        self.min_L = min_L

    def __call__(self, batch):
        return self.collateFunction(batch)

    def collateFunction(self, batch):
        """-> padded_lr (B,min_L,W,H), alphas (B,min_L), hrs (B,W,H) or [], hr_maps (B,W,H) or list, names."""
        lr_batch, alpha_batch, hr_batch, hm_batch, isn_batch = [], [], [], [], []
        train_batch = True
        for imageset in batch:
            lrs = imageset["lr"]
            L, H, W = lrs.shape
            if L >= self.min_L:
                lr_batch.append(lrs[:self.min_L])
                alpha_batch.append(torch.ones(self.min_L))
            else:
                lr_batch.append(torch.cat([lrs, torch.zeros(self.min_L - L, H, W)], dim=0))
                alpha_batch.append(torch.cat([torch.ones(L), torch.zeros(self.min_L - L)], dim=0))
            hr = imageset["hr"]
            if train_batch and hr is not None:
                hr_batch.append(hr)
            else:
                train_batch = False
            hm_batch.append(imageset["hr_map"])
            isn_batch.append(imageset["name"])
        padded_lr_batch = torch.stack(lr_batch, dim=0)
        alpha_batch = torch.stack(alpha_batch, dim=0)
        if train_batch:
            hr_batch = torch.stack(hr_batch, dim=0)
            hm_batch = torch.stack(hm_batch, dim=0)
        return padded_lr_batch, alpha_batch, hr_batch, hm_batch, isn_batch
