#!/usr/bin/env python3
"""Time plane_gemm_lds_kernel (C ABI diinn_plane_gemm_nt) against torch.matmul on the same planes.
usage: plane_gemm_time.py [npix] [M] [Nc] [ksplit]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diinn_amd._native as N  # noqa: E402


def main():
    npix = int(sys.argv[1]) if len(sys.argv) > 1 else 589824
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nc = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    ksplit = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    dev = torch.device("cuda:0")
    lib = N.load()
    t = (npix + 31) // 32
    a_t = torch.randn((t, m, 32), device=dev)            # tiled planes (include/diinn_hip.h)
    b_t = torch.randn((t, nc, 32), device=dev)
    a = torch.randn((m, npix), device=dev)               # plain planes for the library GEMM
    b = torch.randn((nc, npix), device=dev)
    part = torch.empty((ksplit, m, nc + 1), device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def ours():
        N.check(lib.diinn_plane_gemm_nt(stream, C.c_void_p(a_t.data_ptr()), m, 0, C.c_void_p(b_t.data_ptr()), nc, 0,
                                        C.c_void_p(part.data_ptr()), m, nc, npix, ksplit, 1), "plane_gemm")

    def blas():
        return a @ b.t()

    for name, fn in (("plane_gemm_kernel", ours), ("torch.matmul", blas)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{name:20s} M={m} Nc={nc} npix={npix} ksplit={ksplit}: {ms:7.3f} ms  {2.0 * m * nc * npix / ms / 1e9:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
