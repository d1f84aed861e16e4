"""Build libdiinn_hip.so (gfx950) in-tree with hipcc.

    python -m diinn_amd.build            # or: python dual-.../build.py

The library is the whole native product: HIP kernels + the C ABI declared in
include/diinn_hip.h.  It is written next to this file so that it travels with
the repo snapshot to the GPU box (``*.so`` is git-ignored, not gpurun-ignored).
hipcc cross-compiles gfx950 code objects without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_NAME = "libdiinn_hip.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)

SOURCES = ["diinn_kernels.hip", "diinn_host.cpp"]
DEPS = SOURCES + ["diinn_layout.h", os.path.join("..", "..", "include", "diinn_hip.h")]

# -ffp-contract=off: the coordinate formulas must round every fp32 op separately
# (diinn_layout.h axis_eval); the kernels spell out fmaf where fusion is wanted.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-x", "c++"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile the shared library if sources are newer; return its path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = find_hipcc()
    tmp = LIB_PATH + ".tmp"
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    k_obj = os.path.join(objdir, "diinn_kernels.o")
    h_obj = os.path.join(objdir, "diinn_host.o")
    cmds = [
        [hipcc, *HIPCC_FLAGS, "-c", os.path.join(CSRC, "diinn_kernels.hip"), "-o", k_obj],
        [hipcc, *HOST_FLAGS, "-c", os.path.join(CSRC, "diinn_host.cpp"), "-o", h_obj],
        [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, k_obj, h_obj],
    ]
    for cmd in cmds:
        if verbose:
            print("[diinn_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
