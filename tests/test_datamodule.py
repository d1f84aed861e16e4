"""Data side of the callers: datasets and data module in the reference's batch format
(srdata.py, sr_datamodule.py), on small synthetic PNG folders."""
import os
import random

import numpy as np
import pytest
import torch

import diinn_amd.datamodule as DM
from diinn_amd.metrics import resize_fn


def _write_png(path, h, w, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(yy * 3 + seed * 17) % 256, (xx * 5 + seed * 29) % 256, rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(img).save(path)
    return torch.from_numpy(img).permute(2, 0, 1)


@pytest.fixture(scope="module")
def data_root(tmp_path_factory):
    root = tmp_path_factory.mktemp("data")
    imgs = {}
    for i in range(5):
        imgs[f"{i:04d}"] = _write_png(str(root / "DIV2K" / "DIV2K_train_HR" / f"{i:04d}.png"), 60 + 4 * i, 72 + 2 * i, i)
    for i in range(2):
        hr = _write_png(str(root / "benchmark" / "Set5" / "HR" / f"img{i}.png"), 48, 40, 10 + i)
        imgs[f"img{i}"] = hr
        for s in (2, 4):
            from PIL import Image
            lr = resize_fn(hr, (48 // s, 40 // s))
            p = root / "benchmark" / "Set5" / "LR_bicubic" / f"X{s}" / f"img{i}x{s}.png"
            os.makedirs(p.parent, exist_ok=True)
            Image.fromarray(lr.permute(1, 2, 0).numpy()).save(str(p))
    return str(root), imgs


def test_read_rgb_roundtrip(data_root):
    root, imgs = data_root
    got = DM.read_rgb(os.path.join(root, "DIV2K", "DIV2K_train_HR", "0002.png"))
    assert got.dtype == torch.uint8 and torch.equal(got, imgs["0002"])


def test_downsample_dataset_patches(data_root):
    """Training mode: random HR crop of round(patch*scale), LR = antialiased bicubic resize of that crop,
    same flips on both, values in [0,1], one entry per scale (srdata.py:181-236)."""
    root, imgs = data_root
    ds = DM.SRDataDownsample(root=root, name="DIV2K", split="train", scales=[2, 2.5, 3], patch_size=12, augment=False)
    assert len(ds) == 5
    random.seed(3)
    sample = ds[1]
    assert list(sample.keys()) == [2, 2.5, 3]
    for s, (lr, hr, name) in sample.items():
        side = round(12 * s)
        assert lr.shape == (3, 12, 12) and hr.shape == (3, side, side) and name == "0001"
        assert lr.dtype == torch.float32 and 0.0 <= float(lr.min()) and float(hr.max()) <= 1.0
        crop = (hr * 255).round().to(torch.uint8)
        assert torch.equal((lr * 255).round().to(torch.uint8), resize_fn(crop, (12, 12)))
        # the crop really is a window of the stored image
        full = imgs["0001"]
        found = any(torch.equal(full[:, t:t + side, l:l + side], crop)
                    for t in range(full.shape[1] - side + 1) for l in range(full.shape[2] - side + 1))
        assert found
    # same seed -> same patches; random draws in the reference's order (top, left per scale)
    random.seed(3)
    again = ds[1]
    assert all(torch.equal(again[s][1], sample[s][1]) for s in sample)
    random.seed(3)
    tops = []
    full = imgs["0001"]
    for s in (2, 2.5, 3):
        side = round(12 * s)
        t = random.randrange(0, full.shape[1] - side + 1)
        l = random.randrange(0, full.shape[2] - side + 1)
        assert torch.equal((sample[s][1] * 255).round().to(torch.uint8), full[:, t:t + side, l:l + side])


def test_downsample_dataset_whole_image_and_augment(data_root):
    root, imgs = data_root
    ds = DM.SRDataDownsample(root=root, name="DIV2K", split="train", scales=[3.5], patch_size=0, augment=False)
    lr, hr, name = ds[4][3.5]
    h, w = imgs["0004"].shape[1:]
    assert hr.shape == (3, h, w) and lr.shape == (3, round(h / 3.5), round(w / 3.5))
    ds_aug = DM.SRDataDownsample(root=root, name="DIV2K", split="train", scales=[2], patch_size=10, augment=True)
    for seed in range(6):                                  # whatever flips are drawn, lr stays the downsample of hr
        random.seed(seed)
        lr, hr, _ = ds_aug[0][2]
        assert lr.shape == (3, 10, 10) and hr.shape == (3, 20, 20)
        ref = resize_fn((hr * 255).round().to(torch.uint8), (10, 10))
        # flips commute with the resize only up to rounding of the symmetric kernel: compare loosely
        assert float(((lr * 255) - ref.float()).abs().max()) <= 1.0


def test_datamodule_batches_feed_the_module(data_root):
    """SRDataModule.setup + loaders produce the {scale: (lr, hr, name)} batches SRLitModule.step consumes."""
    root, _ = data_root
    dm = DM.SRDataModule(root=root, trainsets=[("DIV2K", "train")], trainsets_repeat=3,
                         testsets=[("benchmark", "Set5")], batch_size=2, train_scales=[2, 3], test_scales=[2, 2.5],
                         patch_size=8, num_workers=0)
    dm.setup()
    assert len(dm.data_train) == 5 * 3 and len(dm.data_test) == 1 and len(dm.data_test[0]) == 2
    assert len(dm.data_val) == 0                           # DIV2K images 800-899 are held out; none in the toy folder
    batch = next(iter(dm.train_dataloader()))
    assert sorted(batch.keys()) == [2, 3]
    lr, hr, names = batch[3]
    assert lr.shape == (2, 3, 8, 8) and hr.shape == (2, 3, 24, 24) and len(names) == 2
    tb = next(iter(dm.test_dataloader()[0]))
    assert tb[2.5][0].shape == (1, 3, round(48 / 2.5), round(40 / 2.5)) and tb[2.5][1].shape == (1, 3, 48, 40)
