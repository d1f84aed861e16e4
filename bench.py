#!/usr/bin/env python3
"""bench.py -- decoded Mpixels/s of the MI355X DIINN implicit-decoder path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "c2"): 256x256 LR encoder features, x4 decode
-> 1024x1024 HR, fp32, synthetic features (seeded N(0,1)) and synthetic weights
drawn from the reference decoder's default-init distribution (synth.py).
A *step* is one full pass of the hot path over one batch: P precompute (hoisted
3x3 conv) + fused decode kernel, inputs already resident in HBM.
N GPUs (weak scaling): the LR map grows to (256*N) x 256 and the HR grid
(1024*N) x 1024 is sharded into N row bands, one per rank; each step first hands
every rank its feature rows (+1-row halo) point-to-point from rank 0 over
RCCL/xGMI, then decodes its band.  Outputs stay sharded (no gather).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LR = 256           # per-GPU LR band height and LR width
SCALE = 4
FLOP_DECODE_PER_PX = 789_504.0        # 3 stacked 512x256 layers + Q0 + head, 2*MAC
FLOP_P_PER_CELL = 1_179_648.0         # hoisted 3x3 conv 64 -> 1024, 2*MAC
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (only for --compute bf16)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sin", choices=["accurate", "hw", "hw_reduced"], default=os.environ.get("DIINN_SIN", "default"),
                    help="sine evaluation of the synthesis branch (default: the library default, hw_reduced)")
    ap.add_argument("--compute", choices=["f32", "bf16", "bf16_full"], default="f32",
                    help="arithmetic of the per-pixel layers; f32 is the reference's precision and the only "
                         "valid headline (bf16 is BASELINE config 5's optional path, 2e-3 relative)")
    ap.add_argument("--dist-mode", choices=["halo", "bcast"], default="halo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def effective_cores() -> int:
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but limits the container to a quota of 16)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(sd, feat_np, size):
    """Reference-faithful CPU decode (oracle/) timed on this host, bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import diinn_oracle as orc
    cores = effective_cores()
    torch.set_num_threads(cores)
    hu, wu = size
    # probe: 64 HR rows; then the reported sample is sized for roughly 10-20 s of CPU work
    t0 = time.perf_counter()
    orc.decode_reference_form(sd, feat_np, size, 30000, row_range=(0, 64))
    t_probe = time.perf_counter() - t0
    rows = int(min(hu, max(64, (12.0 / max(t_probe, 1e-3)) * 64)))
    rows -= rows % 64
    rows = max(rows, 64)
    t0 = time.perf_counter()
    orc.decode_reference_form(sd, feat_np, size, 30000, row_range=(0, rows))
    t = time.perf_counter() - t0
    return {
        "value": round(rows * wu / t / 1e6, 5),
        "unit": "Mpixels/s",
        "cores": cores,
        "kind": "port",
        "sample": f"c2 workload, HR rows 0..{rows} of {hu} ({rows * wu} px), reference-form torch-CPU oracle "
                  f"(unfold -> nearest-exact -> 9 conv1x1 + cat + sin, bsize=30000), {t:.2f} s",
    }


def load_traffic():
    """HBM bytes per decode-kernel launch from the committed rocprofv3 --pmc summary (or None)."""
    p = os.path.join(ROOT, "profiles", "decode_kernel_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the DIINN decode path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run: one rank per GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()           # first RCCL call is a plain collective on every rank

    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_amd.sharded as S
    import diinn_amd.synth as synth

    lib = N.load()
    sin_mode = {"accurate": N.SIN_ACCURATE, "hw": N.SIN_HW, "hw_reduced": N.SIN_HW_REDUCED,
                "default": N.SIN_DEFAULT}[args.sin]
    sin_name = {N.SIN_ACCURATE: "accurate", N.SIN_HW: "hw", N.SIN_HW_REDUCED: "hw_reduced"}[sin_mode]
    sd = synth.decoder_state_dict(123)
    packed = D.pack_state_dict(sd).to(dev)

    H, W = LR * world, LR
    HU, WU = H * SCALE, W * SCALE
    shape = (1, 64, H, W)
    gen = torch.Generator(device=dev)
    gen.manual_seed(123)
    feat = torch.randn(shape, device=dev, generator=gen) if rank == 0 else None
    feat_buf = torch.zeros(shape, device=dev) if rank != 0 else None
    workspace = torch.empty(H * W * 1024, device=dev)
    out = torch.zeros((1, 3, HU, WU), device=dev)

    bands = S.all_bands(HU, world)
    y0, y1 = bands[rank]
    need = [S.feature_rows_for_band(H, D.lr_rows_for_band(H, HU, WU, a, b)) for (a, b) in bands]
    r0, r1 = D.lr_rows_for_band(H, HU, WU, y0, y1)
    stream = torch.cuda.current_stream().cuda_stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        local = feat
        if world > 1:
            local = S.distribute_features(feat, shape, need, src=0, mode=args.dist_mode, device=dev, buf=feat_buf)
        N.check(lib.diinn_precompute_P_ex(C.c_void_p(stream), C.c_void_p(local.data_ptr()), C.c_void_p(packed.data_ptr()),
                                          C.c_void_p(workspace.data_ptr()), 1, H, W, r0, r1, N.COMPUTE[args.compute]),
                "diinn_precompute_P_ex")
        if i is not None:
            ev[i][0].record()
        N.check(lib.diinn_decode_band_ex(C.c_void_p(stream), C.c_void_p(workspace.data_ptr()),
                                         C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr()),
                                         1, H, W, HU, WU, y0, y1, sin_mode, N.COMPUTE[args.compute]),
                "diinn_decode_band_ex")
        if i is not None:
            ev[i][1].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: decode_kernel, HIP-event duration on the launch stream (rank 0's band)
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    px_launch = (y1 - y0) * WU
    achieved = FLOP_DECODE_PER_PX * px_launch / (k_ms * 1e-3) / 1e12

    if rank == 0:
        total_px = HU * WU
        ms_per_step = elapsed / args.steps * 1e3
        res = {
            "metric": "decoded Mpixels/sec (DIINN implicit decoder, x4 on 256^2 LR)",
            "value": round(total_px * args.steps / elapsed / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.compute,
            "data": "synthetic",
            "config": {
                "workload": f"c2: {LR}x{LR} LR encoder features per GPU, x{SCALE} decode -> "
                            f"{HU}x{WU} HR total ({world} row band(s) of {LR * SCALE}x{WU}), B=1, mode=3",
                "lr": [H, W], "hr": [HU, WU], "sin": sin_name,
                "parallelism": f"hr-row-bands x{world}" + (f" ({args.dist_mode} feature hand-off)" if world > 1 else ""),
                "step": "feature hand-off (N>1) + precompute_P + decode_kernel",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "decode_kernel" if args.compute == "f32" else "decode_bf16x2_kernel",
                "achieved": round(achieved, 3),
                "peak": PEAK_F32_MFMA_TFLOPS if args.compute == "f32" else PEAK_BF16_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / (PEAK_F32_MFMA_TFLOPS if args.compute == "f32" else PEAK_BF16_MFMA_TFLOPS), 4),
                "kernel_ms": round(k_ms, 4),
                "flop_per_launch": FLOP_DECODE_PER_PX * px_launch,
                "traffic": load_traffic() if args.compute == "f32" else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, feat.cpu().numpy(), (HU, WU))
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
