"""Data side of the callers: the reference's datasets and data module, so that
``SRLitModule.step`` / ``test_step`` receive batches in the reference's format
``{scale: (lr [B,3,h,w], hr [B,3,H,W], filename)}`` with values in [0, 1].

Behaviour follows /root/reference/src/datamodules/components/srdata.py and sr_datamodule.py:

  SRDataDownsample   HR images only; the LR input is produced on the fly by antialiased bicubic
                     resizing (srdata.py:163-236).  ``patch_size`` > 0: a random HR crop of
                     round(patch_size*scale) pixels resized to patch_size (training); 0: the whole
                     image, LR size round(H/scale) x round(W/scale) (validation / test, any real scale).
  SRDataModule       the train / val / test split and loaders of sr_datamodule.py:50-167 without the
                     Lightning base class (pytorch_lightning is not part of the target image): DIV2K
                     images 0-799 train (repeated), 800-899 validate / test, benchmark sets in full.

Images are decoded with PIL (torchvision.io, which the reference uses, is absent here); decoding an
8-bit RGB PNG gives the same uint8 tensor either way.  Random choices use the ``random`` module in the
reference's order (crop top, crop left, then hflip, vflip, transpose draws), so seeding reproduces
the reference's patches.

Not rebuilt on purpose: the reference's paired-folder dataset with its pickle cache (srdata.py:43-161, ``SRData``) and
``Rotation90`` (sr_datamodule.py:12-20) -- nothing in the reference instantiates either.
"""
from __future__ import annotations

import glob
import os
import random
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset, Subset

from .metrics import resize_fn

# directory names below <root>/<dataset name>/   (srdata.py:11-41)
DATASET_DIR_STRUCTURE: Dict[str, Dict[str, Dict[str, str]]] = {
    "DIV2K": {
        "train": {"hr_dir": "DIV2K_train_HR", "lr_dir": "DIV2K_train_LR_bicubic"},
        "test": {"hr_dir": "DIV2K_test_HR", "lr_dir": "DIV2K_test_LR_bicubic"},
    },
    "benchmark": {
        name: {"hr_dir": f"{name}/HR", "lr_dir": f"{name}/LR_bicubic"} for name in ("B100", "Set5", "Set14", "Urban100")
    },
}


def read_rgb(path: str) -> torch.Tensor:
    """Image file -> uint8 tensor [3,H,W] (what torchvision.io.read_image(path, ImageReadMode.RGB) returns)."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()


def _draw_augmentation():
    """Three coin flips in the reference's order; returns a function applying the same flips to lr and hr."""
    flip_rows = random.random() < 0.5
    flip_cols = random.random() < 0.5
    swap_axes = random.random() < 0.5

    def apply(x: torch.Tensor) -> torch.Tensor:
        if flip_rows:
            x = x.flip(-2)
        if flip_cols:
            x = x.flip(-1)
        if swap_axes:
            x = x.transpose(-2, -1)
        return x

    return apply


class _FolderDataset(Dataset):
    def __init__(self, root, name, split, file_ext, scales, patch_size, augment):
        self.file_ext = file_ext
        self.scales = list(scales)
        self.patch_size = patch_size
        self.augment = augment
        layout = DATASET_DIR_STRUCTURE[name][split]
        self.dataset_dir = Path(root) / name
        self.hr_dir = self.dataset_dir / layout["hr_dir"]
        self.lr_dir = self.dataset_dir / layout["lr_dir"]
        self.names_hr: List[str] = sorted(glob.glob(os.path.join(str(self.hr_dir), "*" + file_ext)))

    def __len__(self) -> int:
        return len(self.names_hr)

    @staticmethod
    def _stem(path: str) -> str:
        return os.path.splitext(os.path.basename(path))[0]


class SRDataDownsample(_FolderDataset):
    """HR folder -> {scale: (lr, hr, filename)}; LR made by antialiased bicubic resize (srdata.py:168-236)."""

    def __init__(self, root: str = "./data/", name: str = "DIV2K", split: str = "train", file_ext: str = ".png",
                 scales: Sequence[float] = (2,), patch_size: int = 96, augment: bool = True):
        super().__init__(root, name, split, file_ext, scales, patch_size, augment)

    def get_patch(self, hr: torch.Tensor, scale: float, patch_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        if patch_size == 0:                                       # whole image
            size = (round(hr.shape[-2] / scale), round(hr.shape[-1] / scale))
            return resize_fn(hr, size), hr
        side = round(patch_size * scale)
        top = random.randrange(0, hr.shape[-2] - side + 1)
        left = random.randrange(0, hr.shape[-1] - side + 1)
        crop = hr[:, top:top + side, left:left + side]
        return resize_fn(crop, (patch_size, patch_size)), crop

    def __getitem__(self, idx: int):
        sample = {}
        path = self.names_hr[idx]
        for scale in self.scales:
            hr = read_rgb(path)
            lr, hr = self.get_patch(hr, scale, self.patch_size)
            if self.augment:
                aug = _draw_augmentation()
                lr, hr = aug(lr), aug(hr)
            sample[scale] = (lr.float() / 255.0, hr.float() / 255.0, self._stem(path))
        return sample


class SRDataModule:
    """Train / val / test datasets and loaders of the reference (sr_datamodule.py:22-167)."""

    DIV2K_TRAIN = range(0, 800)
    DIV2K_HELD_OUT = range(800, 900)

    def __init__(self, root: str = "./data/", trainsets: Sequence[Tuple[str, str]] = (("DIV2K", "train"),),
                 trainsets_repeat: int = 20,
                 testsets: Sequence[Tuple[str, str]] = (("DIV2K", "train"), ("benchmark", "B100"), ("benchmark", "Set5"),
                                                        ("benchmark", "Set14"), ("benchmark", "Urban100")),
                 batch_size: int = 64, train_scales: Sequence[float] = (2, 3, 4),
                 test_scales: Sequence[float] = (2, 2.5, 3, 3.5, 4, 6, 8, 10, 15, 20),
                 patch_size: int = 192, num_workers: int = 16, pin_memory: bool = False):
        self.hparams = SimpleNamespace(root=root, trainsets=list(trainsets), trainsets_repeat=trainsets_repeat,
                                       testsets=list(testsets), batch_size=batch_size, train_scales=list(train_scales),
                                       test_scales=list(test_scales), patch_size=patch_size, num_workers=num_workers,
                                       pin_memory=pin_memory)
        self.data_train: Optional[Dataset] = None
        self.data_val: Optional[Dataset] = None
        self.data_test: Optional[List[Dataset]] = None

    def _dataset(self, name, split, scales, patch_size, augment, div2k_indices):
        ds = SRDataDownsample(root=self.hparams.root, name=name, split=split, scales=scales,
                              patch_size=patch_size, augment=augment)
        if name == "DIV2K":                                       # DIV2K train folder: 0-799 train, 800-899 held out
            ds = Subset(ds, [i for i in div2k_indices if i < len(ds)])
        return ds

    def setup(self, stage: Optional[str] = None) -> None:
        if self.data_train is not None or self.data_test is not None:
            return
        hp = self.hparams
        if hp.trainsets:                                          # evaluation-only callers pass no training set
            train = ConcatDataset([self._dataset(n, s, hp.train_scales, hp.patch_size, True, self.DIV2K_TRAIN)
                                   for n, s in hp.trainsets])
            self.data_train = ConcatDataset([train] * hp.trainsets_repeat)
        self.data_test = [self._dataset(n, s, hp.test_scales, 0, False, self.DIV2K_HELD_OUT) for n, s in hp.testsets]
        self.data_val = self._dataset("DIV2K", "train", hp.train_scales, 0, False, self.DIV2K_HELD_OUT)

    def _loader(self, data, batch_size, shuffle):
        return DataLoader(dataset=data, batch_size=batch_size, num_workers=self.hparams.num_workers,
                          pin_memory=self.hparams.pin_memory, shuffle=shuffle)

    def train_dataloader(self) -> DataLoader:
        return self._loader(self.data_train, self.hparams.batch_size, True)

    def val_dataloader(self) -> DataLoader:
        return self._loader(self.data_val, 1, False)

    def test_dataloader(self) -> List[DataLoader]:
        return [self._loader(d, 1, False) for d in self.data_test]
