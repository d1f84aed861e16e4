#!/usr/bin/env python3
"""Per-layer timing of diinn_conv_ksplit on one MI355X: every (Cin, taps) shape of the RDN trunk at a given map size,
each launched alone in a loop -- shows which layers of the encoder sit furthest below the fp32 MFMA roof.
usage: enc_layer_time.py [SIZE=256] [--wino | --wino4 | --t16] [--relu]   (--t16: the small-map kernel, each shape checked against F.conv2d)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd.modules as M  # noqa: E402
from diinn_amd import _native  # noqa: E402

PEAK = 157.3e12


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lr = int(args[0]) if args else 256
    wino4 = "--wino4" in sys.argv                              # 3x3 layers on diinn_conv_wino4 (F(4x4,3x3))
    wino = "--wino" in sys.argv or wino4                                # 3x3 layers on diinn_conv_wino (% of peak = direct-conv flops / time)
    t16 = "--t16" in sys.argv                                  # 3x3 layers on diinn_conv_t16 (small maps)
    dev = torch.device("cuda:0")
    lib = _native.load()
    hw = lr * lr
    buf = torch.randn(1, 1024 + 64, lr, lr, device=dev) * 0.1
    if "--relu" in sys.argv:                                   # post-ReLU statistics (half zeros), as inside the trunk
        buf.clamp_(min=0)
    bias = torch.zeros(64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    total = 0.0
    shapes = [(64 * k, 9, 16 if k > 1 else 18) for k in range(1, 9)] + [(576, 1, 16), (1024, 1, 1)]
    for cin, taps, count in shapes:
        k = 3 if taps == 9 else 1
        wt = torch.randn(64, cin, k, k) * 0.01
        w = ((M.pack_conv_wino4(wt) if wino4 else M.pack_conv_wino(wt)) if wino and taps == 9 else M.pack_conv_ksplit(wt)).to(dev)
        out = buf[:, 1024:]

        def run():
            if t16 and taps == 1 and cin <= 640:
                _native.check(lib.diinn_conv1x1_t16(stream, C.c_void_p(buf.data_ptr()), (1024 + 64) * hw, cin,
                                                    C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0,
                                                    C.c_void_p(out.data_ptr()), (1024 + 64) * hw, None, 0, 1, 1, lr, lr), "conv")
                return
            if t16 and taps == 9:
                _native.check(lib.diinn_conv_t16(stream, C.c_void_p(buf.data_ptr()), (1024 + 64) * hw, cin,
                                                 C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0,
                                                 C.c_void_p(out.data_ptr()), (1024 + 64) * hw, 1, 1, lr, lr), "conv")
                return
            if wino and taps == 9:
                _native.check((lib.diinn_conv_wino4 if wino4 else lib.diinn_conv_wino)(stream, C.c_void_p(buf.data_ptr()), (1024 + 64) * hw, cin,
                                                  C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0,
                                                  C.c_void_p(out.data_ptr()), (1024 + 64) * hw, 1, 1, lr, lr), "conv")
                return
            _native.check(lib.diinn_conv_ksplit(stream, C.c_void_p(buf.data_ptr()), (1024 + 64) * hw, cin, taps,
                                                C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0,
                                                C.c_void_p(out.data_ptr()), (1024 + 64) * hw, None, 0, 1, 1, lr, lr), "conv")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        if t16 and (taps == 9 or cin <= 640):
            want = torch.relu(torch.nn.functional.conv2d(buf[:, :cin].double(), wt.to(dev).double(), padding=k // 2))
            err = float((out.double() - want).abs().max() / want.abs().max())
            assert err < 2e-6, f"Cin {cin}: relative error {err:.2e}"
        fl = 2.0 * 64 * cin * taps * hw
        total += ms * count
        print(f"Cin {cin:5d} taps {taps}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  {100*fl/ms/1e-3/PEAK:5.1f} %  x{count} = {ms*count:6.3f} ms", flush=True)
    print(f"sum over the trunk's 147 layers: {total:.2f} ms")


if __name__ == "__main__":
    main()
