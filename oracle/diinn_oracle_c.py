"""ctypes front-end of the plain-C oracle (oracle/diinn_oracle_c.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing in the product).  Build with `make -C oracle` (done by
``__graft_entry__.build()``)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libdiinn_oracle_c.so")
_lib = None

ORDER = ["K.0.0.weight", "K.0.0.bias", "K.1.0.weight", "K.1.0.bias", "K.2.0.weight", "K.2.0.bias",
         "K.3.0.weight", "K.3.0.bias", "Q.0.0.weight", "Q.0.0.bias", "Q.1.0.weight", "Q.1.0.bias",
         "Q.2.0.weight", "Q.2.0.bias", "Q.3.0.weight", "Q.3.0.bias", "last_layer.weight", "last_layer.bias"]


def load():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "diinn_oracle_c.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
        _lib.diinn_oracle_decode.restype = C.c_int
    return _lib


def axis_tables(n_in: int, n_out: int, small_output: bool = False):
    lib = load()
    idx = np.empty(n_out, np.int32)
    rel = np.empty(n_out, np.float32)
    lib.diinn_oracle_axis(C.c_int(n_in), C.c_int(n_out), C.c_int(int(small_output)),
                          idx.ctypes.data_as(C.c_void_p), rel.ctypes.data_as(C.c_void_p))
    return idx, rel


def decode(sd, feat, size, row_range=None) -> np.ndarray:
    lib = load()
    feat = np.ascontiguousarray(feat, np.float32)
    b, c, h, w = feat.shape
    hu, wu = int(size[0]), int(size[1])
    y0, y1 = (0, hu) if row_range is None else row_range
    arrs = [np.ascontiguousarray(np.asarray(sd[k], np.float32).reshape(-1)) for k in ORDER]
    ptrs = (C.c_void_p * 18)(*[a.ctypes.data_as(C.c_void_p) for a in arrs])
    out = np.empty((b, 3, y1 - y0, wu), np.float32)
    st = lib.diinn_oracle_decode(feat.ctypes.data_as(C.c_void_p), b, h, w, hu, wu, ptrs,
                                 out.ctypes.data_as(C.c_void_p), y0, y1)
    if st != 0:
        raise RuntimeError("diinn_oracle_decode failed")
    return out
