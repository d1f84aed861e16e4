#!/usr/bin/env python3
"""train.py on the MI355X path: fit DIINN with the reference's configuration file.

The reference's entry point is a LightningCLI (train.py:7-10: ``python train.py fit -c configs/default.yaml``);
pytorch_lightning is not part of the target image, so this is the plain loop around the same hooks:
``SRDataModule`` (batches ``{scale: (lr, hr, name)}``), ``SRLitModule.training_step`` (decoder under
autograd: HIP forward with saved planes + HIP backward), ``configure_optimizers`` (Adam + StepLR per
epoch, sr_module.py:185-194), ``validation_step`` (val/loss, val/psnr_x{scale}), and a ``last.ckpt`` with
the ``state_dict`` / ``hyper_parameters`` entries ``SRLitModule.load_from_checkpoint`` reads.

    python scripts/train.py fit -c /path/to/configs/default.yaml [--max_epochs N] [--limit_train_batches K]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train.py fit -c ...

Read from the YAML (reference configs/default.yaml): ``seed_everything``, ``trainer.max_epochs``,
``trainer.default_root_dir``, ``model.init_args`` (arch, mode, init_q, lr, lr_gamma, lr_step,
eval_bsize) and ``data.init_args`` (SRDataModule arguments).  Under torch.distributed.run every rank
takes one GPU, gradients are averaged by DistributedDataParallel over RCCL (the reference's
``strategy: ddp``), and the training set is split with a DistributedSampler.
"""
import os
import random
import sys
import time
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import yaml  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402
from torch.utils.data.distributed import DistributedSampler  # noqa: E402

from diinn_amd.datamodule import SRDataModule  # noqa: E402
from diinn_amd.modules import SRLitModule  # noqa: E402


def to_device(batch, dev):
    return {scale: (lr.to(dev, non_blocking=True), hr.to(dev, non_blocking=True), names)
            for scale, (lr, hr, names) in batch.items()}


def parse():
    ap = ArgumentParser()
    ap.add_argument("subcommand", choices=["fit", "validate"])
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--max_epochs", type=int, default=None)
    ap.add_argument("--limit_train_batches", type=int, default=None)
    ap.add_argument("--limit_val_batches", type=int, default=None)
    ap.add_argument("--ckpt_path", default=None, help="resume / evaluate from this checkpoint")
    ap.add_argument("--log_every_n_steps", type=int, default=50)
    return ap.parse_args()


def main():
    args = parse()
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    trainer = cfg.get("trainer", {}) or {}
    model_args = (cfg.get("model", {}) or {}).get("init_args", {}) or {}
    data_args = (cfg.get("data", {}) or {}).get("init_args", {}) or {}
    if "trainsets" in data_args:
        data_args["trainsets"] = [tuple(x) for x in data_args["trainsets"]]
    if "testsets" in data_args:
        data_args["testsets"] = [tuple(x) for x in data_args["testsets"]]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a ROCm GPU: the DIINN decoder has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    seed = cfg.get("seed_everything")
    if seed is not None:
        random.seed(seed + rank)
        np.random.seed(seed + rank)
        torch.manual_seed(seed)                                # same initial weights on every rank

    dm = SRDataModule(**data_args)
    dm.setup()
    model = SRLitModule(**model_args) if args.ckpt_path is None else SRLitModule.load_from_checkpoint(args.ckpt_path)
    model = model.to(dev)
    (optimizer,), (scheduler,) = model.configure_optimizers()
    net = model
    if world > 1:
        # DDP wraps the module whose forward the step calls; SRLitModule.step goes through self.forward
        model.net = torch.nn.parallel.DistributedDataParallel(model.net, device_ids=[local_rank])

    def validate(epoch):
        model.eval()
        sums, count = {}, 0
        for i, batch in enumerate(dm.val_dataloader()):
            if args.limit_val_batches is not None and i >= args.limit_val_batches:
                break
            res = model.validation_step(to_device(batch, dev), i)
            for k, v in res.items():
                sums[k] = sums.get(k, 0.0) + float(v)
            count += 1
        if rank == 0 and count:
            print(f"[epoch {epoch}] " + "  ".join(f"{k}={v / count:.4f}" for k, v in sums.items()), flush=True)

    if args.subcommand == "validate":
        validate(-1)
        return

    max_epochs = args.max_epochs if args.max_epochs is not None else int(trainer.get("max_epochs", 50))
    out_dir = trainer.get("default_root_dir") or "."
    os.makedirs(out_dir, exist_ok=True)
    hp = dm.hparams
    sampler = DistributedSampler(dm.data_train, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None
    loader = DataLoader(dm.data_train, batch_size=hp.batch_size, num_workers=hp.num_workers, pin_memory=hp.pin_memory,
                        shuffle=sampler is None, sampler=sampler, drop_last=world > 1)
    for epoch in range(max_epochs):
        model.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0, seen, running = time.perf_counter(), 0, 0.0
        for i, batch in enumerate(loader):
            if args.limit_train_batches is not None and i >= args.limit_train_batches:
                break
            optimizer.zero_grad(set_to_none=True)
            loss = model.training_step(to_device(batch, dev), i)["loss"]
            loss.backward()
            optimizer.step()
            running += float(loss.detach())
            seen += 1
            if rank == 0 and seen % args.log_every_n_steps == 0:
                dt = time.perf_counter() - t0
                print(f"[epoch {epoch} step {seen}] train/loss={running / seen:.5f}  {dt / seen * 1e3:.0f} ms/step", flush=True)
        scheduler.step()
        if rank == 0:
            print(f"[epoch {epoch}] train/loss={running / max(seen, 1):.5f}  lr={scheduler.get_last_lr()[0]:.2e}", flush=True)
        validate(epoch)
        if rank == 0:
            if world > 1:                                       # unwrap DDP so the keys are net.encoder.* / net.decoder.*
                ddp = model.net
                model.net = ddp.module
                ckpt = net.checkpoint()
                model.net = ddp
            else:
                ckpt = net.checkpoint()
            ckpt["epoch"] = epoch
            torch.save(ckpt, os.path.join(out_dir, "last.ckpt"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
