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
    # the split-bf16 decoder writes the same 8-bit image to within one grey level
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "demo2.py"), "--lr_path", str(lr_png),
                        "--output_size", "50", "61", "--ckpt_path", str(ckpt), "--model_name", "diinn_x3",
                        "--compute", "bf16x3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a = np.asarray(Image.open(out)).astype(np.int32)
    b = np.asarray(Image.open(tmp_path / "diinn_x3" / "diinn_x3_img_50x61.png")).astype(np.int32)
    assert a.shape == b.shape and int(np.abs(a - b).max()) <= 1
    # benchmarks.py: a one-image "Set5"
    hr_dir = tmp_path / "data" / "benchmark" / "Set5" / "HR"
    hr_dir.mkdir(parents=True)
    Image.fromarray(rng.integers(0, 255, (64, 72, 3), dtype=np.uint8)).save(hr_dir / "a.png")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "benchmarks.py"), "--ckpt_path", str(ckpt),
                        "--data_root", str(tmp_path / "data")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Set5/psnr_x4" in r.stdout and "Set5/psnr_x3.14" in r.stdout and "Set5/psnr_x8" in r.stdout
    assert "Set5/ssim_x4" in r.stdout and "Set5/lr_psnr_x4" in r.stdout


def test_train_script_command_line_parses():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train.py"), "--help"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "fit" in r.stdout and "--config" in r.stdout


@pytest.mark.gpu
def test_train_script_fits_and_checkpoints(tmp_path):
    """scripts/train.py fit -c <reference-style yaml> on a toy DIV2K folder: a few optimiser steps through the
    HIP forward/backward, then a last.ckpt that SRLitModule.load_from_checkpoint (and the reference) reads."""
    import yaml
    from PIL import Image
    rng = np.random.default_rng(1)
    hr_dir = tmp_path / "data" / "DIV2K" / "DIV2K_train_HR"
    hr_dir.mkdir(parents=True)
    for i in range(4):
        Image.fromarray(rng.integers(0, 255, (80, 96, 3), dtype=np.uint8)).save(hr_dir / f"{i:04d}.png")
    cfg = {
        "seed_everything": 123,
        "trainer": {"max_epochs": 2, "default_root_dir": str(tmp_path / "run")},
        "model": {"class_path": "src.models.sr_module.SRLitModule",
                  "init_args": {"arch": "diinn", "mode": 3, "init_q": False, "lr": 1e-4, "lr_gamma": 0.5, "lr_step": 1,
                                "eval_bsize": 30000}},
        "data": {"class_path": "src.datamodules.sr_datamodule.SRDataModule",
                 "init_args": {"root": str(tmp_path / "data"), "trainsets": [["DIV2K", "train"]], "trainsets_repeat": 1,
                               "testsets": [["DIV2K", "train"]], "batch_size": 2, "train_scales": [2, 3],
                               "test_scales": [2], "patch_size": 12, "num_workers": 0, "pin_memory": False}},
    }
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train.py"), "fit", "-c", str(cfg_path),
                        "--log_every_n_steps", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "[epoch 1] train/loss=" in r.stdout and "lr=2.50e-05" in r.stdout      # StepLR(step 1, gamma 0.5) after 2 epochs
    ckpt = torch.load(tmp_path / "run" / "last.ckpt", map_location="cpu", weights_only=False)
    assert ckpt["hyper_parameters"]["arch"] == "diinn" and ckpt["epoch"] == 1
    keys = set(ckpt["state_dict"].keys())
    assert "net.decoder.K.3.0.weight" in keys and "net.encoder.SFENet1.weight" in keys and "sub" in keys
    import diinn_amd.modules as M
    model = M.SRLitModule.load_from_checkpoint(str(tmp_path / "run" / "last.ckpt"))
    assert model.hparams.lr_step == 1


@pytest.mark.gpu
def test_demo2_sharded_over_ranks_writes_the_same_image(tmp_path, golden):
    """The same demo2 command line under torch.distributed.run: the HR grid is cut into row bands, one per rank (here
    3 ranks on the one GPU, messages over the host-staged gloo transport), rank 0 runs the encoder and assembles the
    image -- which must be the file the single-process run writes, byte for byte."""
    import socket
    from PIL import Image
    ckpt = tmp_path / "last.ckpt"
    _fake_checkpoint(str(ckpt), golden)
    rng = np.random.default_rng(2)
    lr_png = tmp_path / "img.png"
    Image.fromarray(rng.integers(0, 255, (40, 56, 3), dtype=np.uint8)).save(lr_png)
    demo = os.path.join(ROOT, "scripts", "demo2.py")
    args = ["--lr_path", str(lr_png), "--output_size", "132", "185", "--ckpt_path", str(ckpt)]
    r = subprocess.run([sys.executable, demo, *args, "--model_name", "one"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DIINN_BENCH_ONE_DEVICE="1", DIINN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), demo, *args, "--model_name", "three"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    a = np.asarray(Image.open(tmp_path / "one" / "one_img_132x185.png"))
    b = np.asarray(Image.open(tmp_path / "three" / "three_img_132x185.png"))
    assert a.shape == (132, 185, 3) and np.array_equal(a, b)
