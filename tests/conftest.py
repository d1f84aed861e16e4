"""pytest configuration: ``-m gpu`` tests need a real MI355X; everything else runs on CPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


@pytest.fixture(scope="session", autouse=True)
def _native_library_is_built():
    """The tests exercise the in-tree libdiinn_hip.so; (re)build it when sources are newer.
    hipcc cross-compiles gfx950 without a GPU.  No fallback: a failed build fails the session."""
    import diinn_amd.build as b
    b.build(verbose=False)
    yield


@pytest.fixture(scope="session")
def golden():
    """Outputs/tables produced by the real reference decoder (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "diinn_golden.npz"))


@pytest.fixture(scope="session")
def golden_r4():
    """Round-4 additions (tests/golden/make_golden_r4.py): float64 reference outputs of every case (as a float32
    difference to the fp32 output) and the SIREN-range cases."""
    return np.load(os.path.join(ROOT, "tests", "golden", "diinn_golden_r4.npz"))


def siren_cases(g4):
    """(name, b, h, w, hu, wu, gain, q_gain) of the SIREN-range fixtures."""
    out = []
    for k in g4.files:
        if k.startswith("meta/siren"):
            b, h, w, hu, wu, gain, bs, q0, qh = g4[k]
            out.append((k[5:], int(b), int(h), int(w), int(hu), int(wu), float(gain), (float(q0), float(qh))))
    return out


def golden_cases(g):
    out = []
    for k in g.files:
        if k.startswith("meta/"):
            name = k[5:]
            b, h, w, hu, wu, gain, bs = g[k]
            out.append((name, int(b), int(h), int(w), int(hu), int(wu), float(gain)))
    return out


@pytest.fixture
def knobs():
    """Force kernel-variant choices in-process through the C ABI's test hook (diinn_debug_set) and restore them:
    ``knobs("DIINN_F32_KERNEL", 2)``.  (The environment is read once per process, so setenv would do nothing.)"""
    import diinn_amd._native as N
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = N.debug_get(name)
        N.debug_set(name, value)

    yield set_
    for name, value in saved.items():
        N.debug_set(name, value)
