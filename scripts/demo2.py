#!/usr/bin/env python3
"""demo2 on the MI355X path: given an LR image, super-resolve it to the desired resolution.

Same command line as the reference's demo2.py (reference demo2.py:12-19):
    python scripts/demo2.py --lr_path img.png --output_size 512 768 --ckpt_path last.ckpt \
        [--model_name default_model] [--file_ext .png]
Behaviour kept from the reference (demo2.py:29-41): no (x-0.5)/0.5 normalisation, the output
goes to <dir(lr_path)>/<model_name>/<model_name>_<file>_<H>x<W>.png.  Differences: the model and
the image are moved to the GPU (the reference leaves them on the CPU), and image I/O uses PIL
because torchvision is not part of the target image.

Several GPUs: the same command line under ``python -m torch.distributed.run --nproc-per-node N`` cuts
the HR grid into N row bands, one per GPU (rank 0 runs the encoder and writes the file; DESIGN.md
section 7).  DIINN_DIST_BACKEND=gloo with DIINN_BENCH_ONE_DEVICE=1 is the one-GPU test transport.
"""
import os
import sys
from argparse import ArgumentParser
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from diinn_amd.modules import SRLitModule  # noqa: E402


def read_rgb(path):
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(img).permute(2, 0, 1).unsqueeze(0).contiguous()


def save_rgb(t, path):
    arr = (t[0].clamp(0, 1).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy())
    Image.fromarray(arr).save(path)


def _dist_setup():
    """(rank, world, device) -- a process group when launched by torch.distributed.run with more than one rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, torch.device("cuda:0")
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    local = 0 if os.environ.get("DIINN_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("DIINN_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, dev


@torch.no_grad()
def demo2(args):
    rank, world, dev = _dist_setup()
    if args.model_name == "bicubic":
        model = SRLitModule(arch="bicubic")
    else:
        model = SRLitModule.load_from_checkpoint(args.ckpt_path)
    model = model.to(dev).eval()
    if args.compute != "f32" and args.model_name != "bicubic":
        # (not in the reference's command line) arithmetic of the implicit decoder's per-pixel layers: "bf16x3" = split
        # bf16, held to the fp32 tolerance at ~2.9x the decode speed; "bf16" / "bf16_full" = reduced precision
        model.net.decoder.compute = args.compute
    if args.encoder_split_bf16 and args.model_name != "bicubic" and hasattr(model.net, "encoder"):
        model.net.encoder.hip_split_bf16 = True      # the RDN trunk's 3x3 layers in split bf16 (maps of >= ~180x180 pixels)
    if rank == 0:
        print(args.lr_path)
    filename, _ = os.path.splitext(os.path.basename(args.lr_path))
    lr = read_rgb(args.lr_path).to(dev)
    out_dir = os.path.join(os.path.dirname(args.lr_path) or ".", args.model_name)
    if world > 1 and args.model_name != "bicubic":
        sr = model.forward_sharded(lr, args.output_size, src=0, gather_to=0)      # None away from rank 0
    else:
        sr = model(lr, args.output_size) if rank == 0 else None
    if sr is not None:
        Path(out_dir).mkdir(parents=True, exist_ok=True)
        save_rgb(sr, os.path.join(out_dir, "{}_{}_{}x{}.png".format(args.model_name, filename,
                                                                   args.output_size[0], args.output_size[1])))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--lr_path", type=str, required=True)
    parser.add_argument("--output_size", type=int, nargs="+", required=True)
    parser.add_argument("--ckpt_path", type=str, required=True)
    parser.add_argument("--model_name", type=str, default="default_model")
    parser.add_argument("--file_ext", type=str, default=".png")
    parser.add_argument("--compute", type=str, default="f32", choices=["f32", "bf16x3", "bf16", "bf16_full"])
    parser.add_argument("--encoder_split_bf16", action="store_true")
    demo2(parser.parse_args())
