#!/usr/bin/env python3
"""benchmarks.py on the MI355X path: evaluate a checkpoint on the benchmark test sets.

Same command line as the reference (benchmarks.py:6-9): --ckpt_path, --bicubic_test; same data: the
reference's ``SRDataModule(testsets=[B100, Set5, Set14, Urban100], test_scales=[3.14, 4, 8])``
(benchmarks.py:12) -- here ``diinn_amd.datamodule.SRDataModule`` reading ``<data_root>/benchmark/<set>/HR/*.png``
and making the LR input by antialiased bicubic down-sampling of the uint8 image (srdata.py:163-236).  The
reference then drives ``Trainer.test``; Lightning is not part of the target image, so this loop calls
``SRLitModule.test_step`` itself -- normalise, forward(lr, hr.shape[-2:], eval_bsize=30000), de-normalise and
clamp exactly as sr_module.py:113-125,159-160 -- and averages PSNR, SSIM and LR-PSNR per set and scale
(sr_module.py:167-175; metrics re-implemented in diinn_amd/metrics.py).
"""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from diinn_amd.datamodule import SRDataModule  # noqa: E402
from diinn_amd.modules import SRLitModule  # noqa: E402

TESTSETS = ["B100", "Set5", "Set14", "Urban100"]
TEST_SCALES = [3.14, 4, 8]


@torch.no_grad()
def test(args):
    dev = torch.device("cuda:0")
    model = SRLitModule(arch="bicubic") if args.bicubic_test else SRLitModule.load_from_checkpoint(args.ckpt_path)
    model = model.to(dev).eval()
    present = [name for name in TESTSETS if os.path.isdir(os.path.join(args.data_root, "benchmark", name, "HR"))]
    dm = SRDataModule(root=args.data_root, trainsets=[], testsets=[("benchmark", name) for name in present],
                      test_scales=TEST_SCALES, num_workers=0)
    dm.setup()
    results = {}
    for name, loader in zip(present, dm.test_dataloader()):
        sums, count = {}, 0
        for i, batch in enumerate(loader):
            batch = {s: (lr.to(dev), hr.to(dev), names) for s, (lr, hr, names) in batch.items()}
            res = model.test_step(batch, i, 0)
            for scale, r in res.items():
                for key, short in (("psnr_res", "psnr"), ("ssim_res", "ssim"), ("lr_psnr_res", "lr_psnr")):
                    k = f"{name}/{short}_x{scale}"
                    sums[k] = sums.get(k, 0.0) + float(r[key])
            count += 1
        results.update({k: v / count for k, v in sums.items()} if count else {})
    print(results)
    return results


if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--ckpt_path", type=str)
    parser.add_argument("--bicubic_test", action="store_true")
    parser.add_argument("--data_root", type=str, default="./data/", help="SRDataModule root (holds benchmark/<set>/HR)")
    test(parser.parse_args())
