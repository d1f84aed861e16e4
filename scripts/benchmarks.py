#!/usr/bin/env python3
"""benchmarks.py on the MI355X path: evaluate a checkpoint on the benchmark test sets.

Same command line as the reference (benchmarks.py:6-9): --ckpt_path, --bicubic_test.
The reference drives `Trainer.test` over SRDataModule(B100, Set5, Set14, Urban100) at scales
[3.14, 4, 8] (benchmarks.py:12-18); Lightning and the dataset pipeline are outside this tier, so
this counterpart walks `--data_root/<set>/HR/*.png` itself, makes the LR input by antialiased
bicubic down-sampling (the reference's resize_fn, srdata.py:163-166), and calls
`SRLitModule.test_step` -- normalise, forward(lr, hr.shape[-2:], eval_bsize=30000), de-normalise
and clamp exactly as sr_module.py:113-125,159-160 -- reporting PSNR, SSIM and LR-PSNR per set and
scale (sr_module.py:167-175; metrics re-implemented in diinn_amd/metrics.py).
"""
import glob
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from PIL import Image  # noqa: E402

from diinn_amd.modules import SRLitModule  # noqa: E402

TESTSETS = ["B100", "Set5", "Set14", "Urban100"]
TEST_SCALES = [3.14, 4, 8]


def resize_fn(img, size):
    return F.interpolate(img, size=size, mode="bicubic", align_corners=False, antialias=True)


@torch.no_grad()
def test(args):
    dev = torch.device("cuda:0")
    model = SRLitModule(arch="bicubic") if args.bicubic_test else SRLitModule.load_from_checkpoint(args.ckpt_path)
    model = model.to(dev).eval()
    results = {}
    for name in TESTSETS:
        files = sorted(glob.glob(os.path.join(args.data_root, name, "HR", "*.png")))
        for scale in TEST_SCALES:
            psnrs, ssims, lr_psnrs = [], [], []
            for f in files:
                hr = torch.from_numpy(np.asarray(Image.open(f).convert("RGB"), np.float32) / 255.0)
                hr = hr.permute(2, 0, 1).unsqueeze(0).to(dev)
                lr_size = (round(hr.shape[-2] / scale), round(hr.shape[-1] / scale))
                lr = resize_fn(hr, lr_size).clamp(0, 1)
                res = model.test_step({scale: (lr, hr, os.path.basename(f))}, 0, 0)
                psnrs.append(float(res[scale]["psnr_res"]))
                ssims.append(float(res[scale]["ssim_res"]))
                lr_psnrs.append(float(res[scale]["lr_psnr_res"]))
            if psnrs:
                results[f"{name}/psnr_x{scale}"] = sum(psnrs) / len(psnrs)
                results[f"{name}/ssim_x{scale}"] = sum(ssims) / len(ssims)
                results[f"{name}/lr_psnr_x{scale}"] = sum(lr_psnrs) / len(lr_psnrs)
    print(results)
    return results


if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--ckpt_path", type=str)
    parser.add_argument("--bicubic_test", action="store_true")
    parser.add_argument("--data_root", type=str, default="data/benchmark")
    test(parser.parse_args())
