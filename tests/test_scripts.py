"""The reference's entry-point scripts, re-hosted on the MI355X path: same command
lines as /root/reference/demo2.py:12-19 and benchmarks.py:6-9, run as subprocesses on the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import diinn_amd.synth as synth
from conftest import ROOT


def _fake_checkpoint(path, golden):
    full = json.loads(str(golden["diinn/shapes_json"]))
    sd = {"net." + k: torch.from_numpy(v) for k, v in synth.state_dict_for(full, 123, "diinn.").items()}
    sd["sub"] = torch.full((1, 1, 1, 1), 0.5)
    sd["div"] = torch.full((1, 1, 1, 1), 0.5)
    torch.save({"state_dict": sd, "hyper_parameters": {"arch": "diinn", "mode": 3, "init_q": False, "lr": 1e-4,
                                                       "lr_gamma": 0.5, "lr_step": 10, "eval_bsize": 30000}}, path)


def test_script_command_lines_parse():
    for script, args in (("demo2.py", ["--help"]), ("benchmarks.py", ["--help"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), *args],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "demo2.py"), "--help"],
                         capture_output=True, text=True, timeout=300).stdout
    for flag in ("--lr_path", "--output_size", "--ckpt_path", "--model_name", "--file_ext"):
        assert flag in out                       # reference demo2.py:13-18


@pytest.mark.gpu
def test_demo2_and_benchmarks_run_end_to_end(tmp_path, golden):
    from PIL import Image
    ckpt = tmp_path / "last.ckpt"
    _fake_checkpoint(str(ckpt), golden)
    rng = np.random.default_rng(0)
    lr_png = tmp_path / "img.png"
    Image.fromarray(rng.integers(0, 255, (20, 24, 3), dtype=np.uint8)).save(lr_png)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "demo2.py"), "--lr_path", str(lr_png),
                        "--output_size", "50", "61", "--ckpt_path", str(ckpt), "--model_name", "diinn_hip"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path / "diinn_hip" / "diinn_hip_img_50x61.png"      # reference naming, demo2.py:41
    assert out.exists()
    assert Image.open(out).size == (61, 50)
    # benchmarks.py: a one-image "Set5"
    hr_dir = tmp_path / "data" / "Set5" / "HR"
    hr_dir.mkdir(parents=True)
    Image.fromarray(rng.integers(0, 255, (64, 72, 3), dtype=np.uint8)).save(hr_dir / "a.png")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "benchmarks.py"), "--ckpt_path", str(ckpt),
                        "--data_root", str(tmp_path / "data")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Set5/psnr_x4" in r.stdout and "Set5/psnr_x3.14" in r.stdout and "Set5/psnr_x8" in r.stdout
    assert "Set5/ssim_x4" in r.stdout and "Set5/lr_psnr_x4" in r.stdout
