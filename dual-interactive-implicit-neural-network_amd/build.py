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

# one gfx950 translation unit per kernel family (shared definitions: diinn_device.h) + the host-only half of the ABI
HIP_SOURCES = ["diinn_decode.hip", "diinn_precompute.hip", "diinn_precompute_wino.hip", "diinn_precompute_x3.hip", "diinn_bf16.hip", "diinn_bf16x3.hip", "diinn_training.hip",
               "diinn_baselines.hip", "diinn_encoder.hip", "diinn_winograd.hip", "diinn_winograd4.hip", "diinn_conv_x3.hip", "diinn_conv_t16.hip", "diinn_misc.hip"]
HOST_SOURCES = ["diinn_host.cpp"]
SOURCES = HIP_SOURCES + HOST_SOURCES
# (build.py itself counts as a header: it holds the compiler flags of every translation unit)
DEPS = SOURCES + ["diinn_device.h", "diinn_layout.h", "diinn_knobs.h", os.path.join("..", "..", "include", "diinn_hip.h"),
                  os.path.join("..", "build.py")]

# -ffp-contract=off: the coordinate formulas must round every fp32 op separately
# (diinn_layout.h axis_eval); the kernels spell out fmaf where fusion is wanted.
# NaNs are honoured: -fno-honor-nans / -ffast-math / -ffinite-math-only must NOT be (re)introduced here or per file.
# relu is the NaN-propagating v_maximum3_f32 (diinn_device.h: relu0), so a non-finite feature or weight reaches the
# output as it does in the reference; under -fno-honor-nans the compiler may fold that maximum back to v_max_f32 (NaN ->
# 0) and tests/test_gpu_parity.py::test_nonfinite_* fail.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-x", "c++"]
# per translation unit.  -amdgpu-mfma-vgpr-form lets the register allocator keep MFMA accumulators in VGPRs (the
# activations go to AGPRs instead): decode_kernel's epilogue then reads them without v_accvgpr_read -- 988 -> 801 VALU
# instructions per layer, 5.802 -> 5.783 ms at c2 (same box, order-balanced runs, r03).  The same flag makes precompute_P_wino_kernel 8 %
# SLOWER (0.294 -> 0.319 ms), so it is not a library-wide setting.
# Measured TU by TU (tools/r03_vgprform_ab.sh): encoder trunk unchanged in order-balanced runs (11.83-11.88 either way);
# LIIF 12.73 -> 12.60 but MetaSR 6.50 -> 8.52 in the same TU, training step 16.4 -> 16.6: applied to the decode file and to
# the split-bf16 decode (832 -> 260 accumulator moves per layer copy, 2.022 -> 1.995 ms at c2, order-balanced).
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# Compared in order-balanced runs (tools/r03_ab_abba.sh; a run's position in a sequence biases it by ~0.4 %): no flag 5.802,
# vgpr-form 5.783, vgpr-form + -amdgpu-use-amdgpu-trackers 5.800 (noisy), trackers alone 6.05 ms.
# NOT to be used: -amdgpu-disable-unclustered-high-rp-reschedule produced WRONG results (bench.py's oracle check failed).
PER_FILE_FLAGS = {"diinn_decode.hip": _VGPR_FORM, "diinn_bf16x3.hip": _VGPR_FORM}


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


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB_PATH, jobs: int = 4) -> str:
    """Compile the shared library if sources are newer; return its path.  Objects that are newer than
    every dependency are reused; the translation units compile ``jobs`` at a time.  ``extra_flags`` / ``out``
    serve kernel A/B experiments (tools/build_variant.sh)."""
    if not force and not extra_flags and out == LIB_PATH and not needs_build():
        return LIB_PATH
    hipcc = find_hipcc()
    tmp = out + ".tmp"
    objdir = os.path.join(PKG_DIR, "build") if out == LIB_PATH and not extra_flags else out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, d)) for d in DEPS if not d.endswith((".hip", ".cpp")))
    cmds, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        src_path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(src_path)):
            continue
        flags = [*HIPCC_FLAGS, *PER_FILE_FLAGS.get(src, []), *extra_flags] if src in HIP_SOURCES else HOST_FLAGS
        cmds.append([hipcc, *flags, "-c", src_path, "-o", obj])
    running = []
    for cmd in cmds:
        if verbose:
            print("[diinn_amd.build]", " ".join(cmd), flush=True)
        running.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        if len(running) >= jobs:
            c, pr = running.pop(0)
            if pr.wait():
                raise subprocess.CalledProcessError(pr.returncode, c)
    for c, pr in running:
        if pr.wait():
            raise subprocess.CalledProcessError(pr.returncode, c)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    if verbose:
        print("[diinn_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=CSRC)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
