#!/usr/bin/env python3
"""Decoder training step (forward + backward) timing on one GPU.

  ours   : DecodeMode3Function (HIP forward with saved planes + library-GEMM backward)
  eager  : the reference's op sequence in PyTorch-ROCm eager mode (unfold -> nearest-exact gather ->
           9 conv1x1 + 3 cat + 4 sin under autograd), restated inline (the reference itself does not
           travel to the GPU box)

usage: train_time.py [B] [LR] [SCALE] [--only-ours]     (default 16 48 4: the reference's training patch geometry,
       configs/default.yaml: batch 16, 48x48 LR patches, scales 2-4)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diinn_amd.decoder as D  # noqa: E402
import diinn_amd.synth as synth  # noqa: E402
import diinn_amd.training as T  # noqa: E402


def eager_forward(dec, feat, size, idx_h, idx_w, syn):
    u = F.unfold(feat, 3, padding=1).view(feat.shape[0], 576, feat.shape[2], feat.shape[3])
    x = u[:, :, idx_h][:, :, :, idx_w]
    k = dec.K[0](x)
    q = k * dec.Q[0](syn)
    for i in range(1, 4):
        k = dec.K[i](torch.cat([q, x], dim=1))
        q = k * dec.Q[i](q)
    return dec.last_layer(q)


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def e2e():
    """One optimiser step of the whole network the way the reference trains it (configs/default.yaml:
    batch 16, 48x48 LR patches; SRLitModule.step loops over the scales of the batch, sr_module.py:113-125),
    RDN encoder on PyTorch-ROCm/MIOpen in both cases; decoder on the HIP path vs the eager op sequence."""
    import diinn_amd.modules as M
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = M.SRLitModule(arch="diinn", mode=3, init_q=False).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    lr = torch.rand(16, 3, 48, 48, device=dev)
    batch = {s: (lr, torch.rand(16, 3, 48 * s, 48 * s, device=dev), None) for s in (2, 3, 4)}
    dec = net.net.decoder
    tabs = {}
    for s in batch:
        idx_h, rel_h, idx_w, rel_w, ratio = T.coordinate_tensors(48, 48, 48 * s, 48 * s, dev)
        syn = torch.empty(16, 3, 48 * s, 48 * s, device=dev)
        syn[:, 0] = rel_h[None, :, None]
        syn[:, 1] = rel_w[None, None, :]
        syn[:, 2] = ratio
        tabs[s] = (idx_h, idx_w, syn)

    def step_ours():
        opt.zero_grad(set_to_none=True)
        loss, _ = net.step(batch)
        loss.backward()
        opt.step()

    def step_eager():
        opt.zero_grad(set_to_none=True)
        loss = 0
        for s, (x, hr, _) in batch.items():
            feat = net.net.encoder((x - net.sub) / net.div)
            pred = eager_forward(dec, feat, hr.shape[-2:], tabs[s][0], tabs[s][1], tabs[s][2])
            loss = loss + net.criterion(pred, (hr - net.sub) / net.div)
        (loss / len(batch)).backward()
        opt.step()

    t_ours = timeit(step_ours, n=5, warm=3)
    print(f"full training step, batch 16 x 48x48 LR, scales 2+3+4 (encoder + decoder + Adam)")
    print(f"  decoder on the HIP path   {t_ours:8.1f} ms")
    t_eager = timeit(step_eager, n=3, warm=2)
    print(f"  decoder as eager ops      {t_eager:8.1f} ms   ({t_eager / t_ours:.2f}x)")
    print(f"  peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


def main():
    only_ours = "--only-ours" in sys.argv          # for profiling: skip the eager comparison
    if "--e2e" in sys.argv:
        return e2e()
    argv = [a for a in sys.argv if not a.startswith("--")]
    b = int(argv[1]) if len(argv) > 1 else 16
    lr = int(argv[2]) if len(argv) > 2 else 48
    sc = int(argv[3]) if len(argv) > 3 else 4
    dev = torch.device("cuda:0")
    hu = wu = lr * sc
    dec = D.ImplicitDecoder(mode=3, init_q=False)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state_dict(123).items()})
    dec = dec.to(dev).train()
    feat = torch.from_numpy(synth.encoder_features(123, b, lr, lr)).to(dev).requires_grad_(True)
    r = torch.randn(b, 3, hu, wu, device=dev)
    idx_h, rel_h, idx_w, rel_w, ratio = T.coordinate_tensors(lr, lr, hu, wu, dev)
    syn = torch.empty(b, 3, hu, wu, device=dev)
    syn[:, 0] = rel_h[None, :, None]
    syn[:, 1] = rel_w[None, None, :]
    syn[:, 2] = ratio

    def ours_fwd():
        with torch.no_grad():
            dec(feat, [hu, wu], 30000)

    def ours_train_fwd():
        return dec(feat, [hu, wu])

    def ours_step():
        dec.zero_grad(set_to_none=True)
        feat.grad = None
        (dec(feat, [hu, wu]) * r).sum().backward()

    def eager_step():
        dec.zero_grad(set_to_none=True)
        feat.grad = None
        (eager_forward(dec, feat, (hu, wu), idx_h, idx_w, syn) * r).sum().backward()

    n = b * hu * wu
    print(f"B={b} LR={lr}x{lr} x{sc} -> {hu}x{wu}: {n} HR pixels")
    print(f"  inference forward (no grad)        {timeit(ours_fwd):8.2f} ms")
    print(f"  training forward (saves planes)    {timeit(ours_train_fwd):8.2f} ms")
    t_ours = timeit(ours_step)
    print(f"  training step fwd+bwd (ours)       {t_ours:8.2f} ms   {n / t_ours / 1e3:.1f} Mpix/s")
    if only_ours:
        return
    y = eager_forward(dec, feat, (hu, wu), idx_h, idx_w, syn)
    with torch.no_grad():
        err = float((y - dec(feat, [hu, wu], 30000)).abs().max())
    del y
    t_eager = timeit(eager_step, n=3, warm=1)
    print(f"  training step fwd+bwd (eager ops)  {t_eager:8.2f} ms   {n / t_eager / 1e3:.1f} Mpix/s   (max|out diff| {err:.1e})")
    print(f"  speedup {t_eager / t_ours:.2f}x;  peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
