#!/usr/bin/env python3
"""Time the fp32 hoisted 3x3 conv (diinn_precompute_P_ex) over map sizes: the Winograd kernel against the direct one
(DIINN_P_KERNEL = 2 / 1 in separate processes).  usage: p_time.py [SIZE ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.synth as synth

dev = torch.device("cuda:0")
lib = N.load()
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for lr in [int(a) for a in sys.argv[1:]] or [16, 24, 48, 64, 96, 128, 181, 256]:
    feat = torch.randn(1, 64, lr, lr, device=dev)
    P = torch.empty(lr * lr * 1024, device=dev)
    run = lambda: N.check(lib.diinn_precompute_P_ex(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                                    C.c_void_p(P.data_ptr()), 1, lr, lr, 0, lr, N.COMPUTE_F32), "P")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{lr:4d}x{lr:<4d} {us:9.1f} us   {1179648.0 * lr * lr / us / 1e6:7.1f} TFLOP/s (direct-convolution FLOPs)", flush=True)
