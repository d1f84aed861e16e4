"""Host side of the MI355X DIINN decode path: the reference's ``ImplicitDecoder``
interface, executed by the gfx950 kernels in libdiinn_hip.so.

Mirrors /root/reference/src/models/components/diinn.py:
  * ``ImplicitDecoder(in_channels=64, hidden_dims=[256]*4, mode=1, init_q=False)``
    registers parameters under the same names and shapes as diinn.py:40-92
    (``K.{i}.0.weight`` ..., ``Q.{i}.0.*``, ``last_layer.*``) so reference
    checkpoints ``load_state_dict`` unchanged;
  * ``forward(x, size, bsize=None)`` has the argument meaning of diinn.py:163-173.

The compute is the HIP path and only the HIP path: CPU tensors, a missing
library or decoder variants the kernels do not cover raise instead of silently
falling back.  PyTorch is used for device memory and the stream; under autograd
(training, mode 3) the forward is the HIP kernel with saved activations and the
backward is library GEMMs over those (training.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _native

IN_CHANNELS = 64
HIDDEN = 256
P_CHANNELS = 1024


class SineAct(nn.Module):
    """sin activation of the synthesis branch (reference: diinn.py:21-26)."""

    def forward(self, x):
        return torch.sin(x)


# ---------------------------------------------------------------------------
# packed weights
# ---------------------------------------------------------------------------
def pack_state_dict(sd, prefix: str = "", mode: int = 3) -> torch.Tensor:
    """Reference-named decoder tensors -> packed host image (1-D fp32 CPU tensor).

    ``sd`` maps ``{prefix}K.0.0.weight`` ... to tensors/arrays in the layouts of
    SURVEY.md App. A.1 (init_q=False).  Modes 2 and 3 share the layout (K.i is [256, 832]: 256
    chained inputs, then the 576 unfolded features); mode 1's K.i is [256, 256] (no feature
    columns, diinn.py:53-60) and is widened with zero feature columns, so P_i = bK_i for i >= 1.
    Calls the C ABI ``diinn_pack_weights`` (include/diinn_hip.h)."""
    lib = _native.load()

    def get(name, shape):
        t = sd[prefix + name]
        a = t.detach().to("cpu", torch.float32).numpy() if isinstance(t, torch.Tensor) else np.asarray(t, np.float32)
        a = np.ascontiguousarray(a.reshape(shape), dtype=np.float32)
        return a

    k0w = get("K.0.0.weight", (HIDDEN, 576)); k0b = get("K.0.0.bias", (HIDDEN,))
    if mode == 1:
        kw = []
        for i in (1, 2, 3):
            wide = np.zeros((HIDDEN, HIDDEN + 576), np.float32)
            wide[:, :HIDDEN] = get(f"K.{i}.0.weight", (HIDDEN, HIDDEN))
            kw.append(wide)
    else:
        kw = [get(f"K.{i}.0.weight", (HIDDEN, HIDDEN + 576)) for i in (1, 2, 3)]
    kb = [get(f"K.{i}.0.bias", (HIDDEN,)) for i in (1, 2, 3)]
    q0w = get("Q.0.0.weight", (HIDDEN, 3)); q0b = get("Q.0.0.bias", (HIDDEN,))
    qw = [get(f"Q.{i}.0.weight", (HIDDEN, HIDDEN)) for i in (1, 2, 3)]
    qb = [get(f"Q.{i}.0.bias", (HIDDEN,)) for i in (1, 2, 3)]
    lw = get("last_layer.weight", (3, HIDDEN)); lb = get("last_layer.bias", (3,))

    packed = np.empty(lib.diinn_packed_weight_floats(), dtype=np.float32)
    f3 = _native._f3
    st = lib.diinn_pack_weights(
        _native.fptr(k0w), _native.fptr(k0b),
        f3(*[_native.fptr(a) for a in kw]), f3(*[_native.fptr(a) for a in kb]),
        _native.fptr(q0w), _native.fptr(q0b),
        f3(*[_native.fptr(a) for a in qw]), f3(*[_native.fptr(a) for a in qb]),
        _native.fptr(lw), _native.fptr(lb), _native.fptr(packed))
    _native.check(st, "diinn_pack_weights")
    return torch.from_numpy(packed)


def axis_tables(n_in: int, n_out: int, small_output: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Host tables (idx int32, rel fp32) from the C ABI ``diinn_make_axis_tables``."""
    lib = _native.load()
    idx = np.empty(n_out, np.int32)
    rel = np.empty(n_out, np.float32)
    st = lib.diinn_make_axis_tables(n_in, n_out, int(small_output),
                                    idx.ctypes.data_as(_native._i32), _native.fptr(rel))
    _native.check(st, "diinn_make_axis_tables")
    return idx, rel


def lr_rows_for_band(h: int, hu: int, wu: int, y0: int, y1: int) -> Tuple[int, int]:
    """LR rows [r0,r1) whose cells HR rows [y0,y1) read (C ABI ``diinn_lr_rows_for_band``)."""
    lib = _native.load()
    r0, r1 = C.c_int(), C.c_int()
    _native.check(lib.diinn_lr_rows_for_band(h, hu, wu, y0, y1, C.byref(r0), C.byref(r1)),
                  "diinn_lr_rows_for_band")
    return r0.value, r1.value


# ---------------------------------------------------------------------------
# functional entry: features -> RGB on the current HIP stream
# ---------------------------------------------------------------------------
def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"diinn_amd: {what} must live on a ROCm GPU (got device {t.device}); the MI355X decode "
            f"path has no CPU implementation")


def decode_features(feat: torch.Tensor, packed: torch.Tensor, size: Sequence[int],
                    out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                    rows: Optional[Tuple[int, int]] = None, sin_mode: int = _native.SIN_DEFAULT,
                    compute: str = "f32", mode: int = 3) -> torch.Tensor:
    """Decode encoder features ``feat`` [B,64,H,W] to RGB [B,3,Hu,Wu].

    ``rows=(y0,y1)`` computes only that HR row band (tile sharding across GPUs);
    the rest of ``out`` is left untouched.  ``workspace`` is the P image
    [B,H,W,1024] fp32 (allocated if None).  ``compute`` = "f32" (reference precision), "bf16" /
    "bf16_full" (bf16 operands in layers 1..3 / also in the hoisted conv, fp32 accumulate; ~2e-3 relative) or
    "bf16x3" (split bf16: hi + lo bf16 operands, three bf16 MFMA products per term; held to f32's 1e-4 bound).  ``mode`` 3 (default) is the
    reference's final model; modes 1 and 2 (packed with ``pack_state_dict(..., mode=...)``) run fp32
    only.  Enqueues two kernels (three for modes 1/2: + the per-cell modulation chain) on the
    current stream; never synchronises."""
    lib = _native.load()
    _require_cuda(feat, "feat")
    _require_cuda(packed, "packed weights")
    if feat.dtype != torch.float32 or feat.dim() != 4 or feat.shape[1] != IN_CHANNELS:
        raise ValueError(f"feat must be fp32 [B,{IN_CHANNELS},H,W], got {feat.dtype} {tuple(feat.shape)}")
    hu, wu = size  # same unpack as the reference (diinn.py:96): raises unless len(size) == 2
    hu, wu = int(hu), int(wu)
    feat = feat.contiguous()
    b, _, h, w = feat.shape
    y0, y1 = (0, hu) if rows is None else (int(rows[0]), int(rows[1]))
    if out is None:
        out = torch.empty((b, 3, hu, wu), dtype=torch.float32, device=feat.device)
    else:
        if out.shape != (b, 3, hu, wu) or out.dtype != torch.float32 or not out.is_contiguous() \
                or out.device != feat.device:
            raise ValueError("out must be a contiguous fp32 [B,3,Hu,Wu] tensor on feat's device")
    need = lib.diinn_workspace_bytes(b, h, w)
    if workspace is None:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=feat.device)
    elif workspace.numel() * workspace.element_size() < need or not workspace.is_contiguous() \
            or workspace.device != feat.device:
        raise ValueError(f"workspace must be a contiguous buffer of >= {need} bytes on feat's device")
    if mode not in (1, 2, 3):
        raise NotImplementedError(f"mode {mode}: the HIP path covers modes 1-3")
    if mode != 3 and compute != "f32":
        raise ValueError("modes 1 and 2 run in fp32 only")
    comp = _native.COMPUTE[compute] if mode == 3 else _native.COMPUTE_F32_QONLY
    with torch.cuda.device(feat.device):
        stream = torch.cuda.current_stream().cuda_stream
        st = lib.diinn_decode_ex(C.c_void_p(stream), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                 C.c_void_p(workspace.data_ptr()), C.c_void_p(out.data_ptr()),
                                 b, h, w, hu, wu, y0, y1, int(sin_mode), comp)
    _native.check(st, "diinn_decode_ex")
    return out


def window_rows(h: int, hu: int, wu: int, y0: int, y1: int) -> Tuple[Tuple[int, int], Tuple[int, int]]:
    """((feat_row0, feat_rows), (p_row0, p_rows)): the LR feature rows (cells + 3x3 halo) and the P rows that
    decoding HR rows [y0,y1) touches (C ABI ``diinn_window_rows``)."""
    lib = _native.load()
    a0, an, r0, rn = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _native.check(lib.diinn_window_rows(h, hu, wu, y0, y1, C.byref(a0), C.byref(an), C.byref(r0), C.byref(rn)),
                  "diinn_window_rows")
    return (a0.value, an.value), (r0.value, rn.value)


def decode_window(feat_win: torch.Tensor, feat_row0: int, full_h: int, packed: torch.Tensor, size: Sequence[int],
                  rows: Tuple[int, int], p_win: Optional[torch.Tensor] = None, out_win: Optional[torch.Tensor] = None,
                  sin_mode: int = _native.SIN_DEFAULT, compute: str = "f32", mode: int = 3) -> torch.Tensor:
    """Decode HR rows ``rows=(y0,y1)`` from band-sized buffers (the multi-GPU row-band unit).

    ``feat_win`` [B,64,fr,W] holds LR rows [feat_row0, feat_row0+fr) of a map of height ``full_h`` and must
    cover the band's cells plus the 3x3 halo (``window_rows``).  ``p_win`` is a workspace of at least
    B*p_rows*W*1024 floats and ``out_win`` [B,3,y1-y0,Wu] the band of the output (both allocated if None).
    Bit-identical to the same rows of ``decode_features`` on the full map.  C ABI: ``diinn_decode_win``."""
    lib = _native.load()
    _require_cuda(feat_win, "feat_win")
    _require_cuda(packed, "packed weights")
    if feat_win.dtype != torch.float32 or feat_win.dim() != 4 or feat_win.shape[1] != IN_CHANNELS \
            or not feat_win.is_contiguous():
        raise ValueError(f"feat_win must be contiguous fp32 [B,{IN_CHANNELS},rows,W], got {feat_win.dtype} "
                         f"{tuple(feat_win.shape)}")
    hu, wu = size
    hu, wu = int(hu), int(wu)
    y0, y1 = int(rows[0]), int(rows[1])
    b, _, fr, w = feat_win.shape
    (_, _), (r0, rn) = window_rows(int(full_h), hu, wu, y0, y1)
    need = b * rn * w * P_CHANNELS
    if p_win is None:
        p_win = torch.empty(need, dtype=torch.float32, device=feat_win.device)
    elif p_win.numel() < need or p_win.dtype != torch.float32 or not p_win.is_contiguous() \
            or p_win.device != feat_win.device:
        raise ValueError(f"p_win must be a contiguous fp32 buffer of >= {need} floats on feat_win's device")
    if out_win is None:
        out_win = torch.empty((b, 3, y1 - y0, wu), dtype=torch.float32, device=feat_win.device)
    elif out_win.shape != (b, 3, y1 - y0, wu) or out_win.dtype != torch.float32 or not out_win.is_contiguous() \
            or out_win.device != feat_win.device:
        raise ValueError("out_win must be a contiguous fp32 [B,3,y1-y0,Wu] tensor on feat_win's device")
    if mode not in (1, 2, 3):
        raise NotImplementedError(f"mode {mode}: the HIP path covers modes 1-3")
    if mode != 3 and compute != "f32":
        raise ValueError("modes 1 and 2 run in fp32 only")
    comp = _native.COMPUTE[compute] if mode == 3 else _native.COMPUTE_F32_QONLY
    with torch.cuda.device(feat_win.device):
        stream = torch.cuda.current_stream().cuda_stream
        st = lib.diinn_decode_win(C.c_void_p(stream), C.c_void_p(feat_win.data_ptr()), int(feat_row0), fr,
                                  C.c_void_p(packed.data_ptr()), C.c_void_p(p_win.data_ptr()), r0, rn,
                                  C.c_void_p(out_win.data_ptr()), y0, y1 - y0,
                                  b, int(full_h), w, hu, wu, y0, y1, int(sin_mode), comp)
    _native.check(st, "diinn_decode_win")
    return out_win


def decode_tile(p_win: torch.Tensor, p_row0: int, shape: Sequence[int], packed: torch.Tensor, size: Sequence[int],
                rows: Tuple[int, int], cols: Tuple[int, int], out: torch.Tensor,
                sin_mode: int = _native.SIN_DEFAULT, compute: str = "f32") -> torch.Tensor:
    """Decode the HR tile rows x cols = [y0,y1) x [x0,x1) from a P window (``diinn_precompute_P_win``'s output: LR rows
    [p_row0, p_row0 + p_rows) of the [B,H,W,1024] image, ``shape`` = (B, H, W) of the full map) INTO ``out``, any fp32
    view of shape [B,3,y1-y0,x1-x0] with unit stride along x -- a tensor of its own or a window of a larger canvas;
    nothing else of the canvas is written.  Bit-identical to the same pixels of ``decode_features``.  The reference's
    analogue is ``batched_step``'s column strips (diinn.py:149-160).  C ABI: ``diinn_decode_tile_win``."""
    lib = _native.load()
    _require_cuda(p_win, "p_win")
    _require_cuda(packed, "packed weights")
    b, h, w = (int(v) for v in shape)
    hu, wu = int(size[0]), int(size[1])
    y0, y1 = int(rows[0]), int(rows[1])
    x0, x1 = int(cols[0]), int(cols[1])
    if out.dtype != torch.float32 or out.dim() != 4 or tuple(out.shape) != (b, 3, y1 - y0, x1 - x0) or out.device != p_win.device:
        raise ValueError(f"out must be an fp32 [B,3,{y1 - y0},{x1 - x0}] view on the P window's device, got {out.dtype} {tuple(out.shape)}")
    if out.stride(3) != 1 and x1 - x0 > 1:
        raise ValueError("out must have unit stride along x")
    if p_win.dtype != torch.float32 or not p_win.is_contiguous() or p_win.numel() % (b * w * P_CHANNELS):
        raise ValueError("p_win must be a contiguous fp32 buffer of B * rows * W * 1024 floats")
    p_rows = p_win.numel() // (b * w * P_CHANNELS)
    with torch.cuda.device(p_win.device):
        stream = torch.cuda.current_stream().cuda_stream
        st = lib.diinn_decode_tile_win(C.c_void_p(stream), C.c_void_p(p_win.data_ptr()), int(p_row0), p_rows,
                                       C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr()),
                                       out.stride(2), out.stride(1), out.stride(0), b, h, w, hu, wu, y0, y1, x0, x1,
                                       int(sin_mode), _native.COMPUTE[compute])
    _native.check(st, "diinn_decode_tile_win")
    return out


# ---------------------------------------------------------------------------
# LIIF comparison decoder (reference liif.py)
# ---------------------------------------------------------------------------
def pack_liif_state_dict(sd, prefix: str = "imnet.") -> torch.Tensor:
    """``imnet`` of the reference LIIF (MLP 580 -> 256 x4 -> 3, liif.py:24, mlp.py) -> the DIINN packed image,
    with the slot mapping documented at ``diinn_liif_decode`` in include/diinn_hip.h: the 576 feature
    columns of layer 0 become the hoisted 3x3 conv, its 4 coordinate columns the Q0 table, layers 2/4/6
    the synthesis slots of the stacked per-pixel layers, layer 8 the RGB head."""
    def get(name, shape):
        t = sd[prefix + name]
        a = t.detach().to("cpu", torch.float32).numpy() if isinstance(t, torch.Tensor) else np.asarray(t, np.float32)
        return np.ascontiguousarray(a.reshape(shape), dtype=np.float32)

    w0 = get("layers.0.weight", (HIDDEN, 580))
    mapped = {
        "K.0.0.weight": w0[:, :576], "K.0.0.bias": get("layers.0.bias", (HIDDEN,)),
        "Q.0.0.weight": w0[:, 576:579], "Q.0.0.bias": w0[:, 579],
        "last_layer.weight": get("layers.8.weight", (3, HIDDEN)), "last_layer.bias": get("layers.8.bias", (3,)),
    }
    for i, layer in ((1, 2), (2, 4), (3, 6)):
        mapped[f"K.{i}.0.weight"] = np.zeros((HIDDEN, HIDDEN + 576), np.float32)
        mapped[f"K.{i}.0.bias"] = np.zeros((HIDDEN,), np.float32)
        mapped[f"Q.{i}.0.weight"] = get(f"layers.{layer}.weight", (HIDDEN, HIDDEN))
        mapped[f"Q.{i}.0.bias"] = get(f"layers.{layer}.bias", (HIDDEN,))
    return pack_state_dict(mapped, mode=3)


def liif_axis_tables(n_in: int, n_out: int, v: int) -> Tuple[np.ndarray, np.ndarray, float]:
    """Host tables (idx int32, rel fp32) and rel_cell of one axis for ensemble shift v (C ABI)."""
    lib = _native.load()
    idx = np.empty(n_out, np.int32)
    rel = np.empty(n_out, np.float32)
    cell = C.c_float()
    _native.check(lib.diinn_liif_make_axis_tables(n_in, n_out, v, idx.ctypes.data_as(_native._i32), _native.fptr(rel),
                                                  C.byref(cell)), "diinn_liif_make_axis_tables")
    return idx, rel, cell.value


def liif_decode_features(feat: torch.Tensor, packed: torch.Tensor, size: Sequence[int],
                         out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LIIF query of every HR pixel: encoder features [B,64,H,W] -> RGB [B,3,Hu,Wu] (liif.py:59-127,148-155)."""
    lib = _native.load()
    _require_cuda(feat, "feat")
    _require_cuda(packed, "packed weights")
    if feat.dtype != torch.float32 or feat.dim() != 4 or feat.shape[1] != IN_CHANNELS:
        raise ValueError(f"feat must be fp32 [B,{IN_CHANNELS},H,W], got {feat.dtype} {tuple(feat.shape)}")
    hu, wu = size
    hu, wu = int(hu), int(wu)
    feat = feat.contiguous()
    b, _, h, w = feat.shape
    if out is None:
        out = torch.empty((b, 3, hu, wu), dtype=torch.float32, device=feat.device)
    need = lib.diinn_workspace_bytes(b, h, w)
    if workspace is None or workspace.numel() * 4 < need or workspace.device != feat.device:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        stream = torch.cuda.current_stream().cuda_stream
        st = lib.diinn_liif_decode(C.c_void_p(stream), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                   C.c_void_p(workspace.data_ptr()), C.c_void_p(out.data_ptr()), b, h, w, hu, wu)
    _native.check(st, "diinn_liif_decode")
    return out


# ---------------------------------------------------------------------------
# MetaSR comparison decoder (reference metasr.py)
# ---------------------------------------------------------------------------
def pack_metasr_state_dict(sd, prefix: str = "imnet.") -> torch.Tensor:
    """``imnet`` of the reference MetaSR (Linear(3,256), ReLU, Linear(256,1728); metasr.py:27-35) -> its packed
    image (C ABI ``diinn_metasr_pack_weights``)."""
    lib = _native.load()

    def get(name, shape):
        t = sd[prefix + name]
        a = t.detach().to("cpu", torch.float32).numpy() if isinstance(t, torch.Tensor) else np.asarray(t, np.float32)
        return np.ascontiguousarray(a.reshape(shape), dtype=np.float32)

    w1, b1 = get("layers.0.weight", (HIDDEN, 3)), get("layers.0.bias", (HIDDEN,))
    w2, b2 = get("layers.2.weight", (1728, HIDDEN)), get("layers.2.bias", (1728,))
    packed = np.empty(lib.diinn_metasr_packed_floats(), dtype=np.float32)
    _native.check(lib.diinn_metasr_pack_weights(_native.fptr(w1), _native.fptr(b1), _native.fptr(w2), _native.fptr(b2),
                                                _native.fptr(packed)), "diinn_metasr_pack_weights")
    return torch.from_numpy(packed)


def metasr_axis_tables(n_in: int, n_out: int) -> Tuple[np.ndarray, np.ndarray, float]:
    lib = _native.load()
    idx = np.empty(n_out, np.int32)
    rel = np.empty(n_out, np.float32)
    r_rev = C.c_float()
    _native.check(lib.diinn_metasr_make_axis_tables(n_in, n_out, idx.ctypes.data_as(_native._i32), _native.fptr(rel),
                                                    C.byref(r_rev)), "diinn_metasr_make_axis_tables")
    return idx, rel, r_rev.value


def metasr_decode_features(feat: torch.Tensor, packed: torch.Tensor, size: Sequence[int],
                           out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MetaSR query of every HR pixel: encoder features [B,64,H,W] -> RGB [B,3,Hu,Wu] (metasr.py:70-104,125-135)."""
    lib = _native.load()
    _require_cuda(feat, "feat")
    _require_cuda(packed, "packed weights")
    if feat.dtype != torch.float32 or feat.dim() != 4 or feat.shape[1] != IN_CHANNELS:
        raise ValueError(f"feat must be fp32 [B,{IN_CHANNELS},H,W], got {feat.dtype} {tuple(feat.shape)}")
    hu, wu = size
    hu, wu = int(hu), int(wu)
    feat = feat.contiguous()
    b, _, h, w = feat.shape
    if out is None:
        out = torch.empty((b, 3, hu, wu), dtype=torch.float32, device=feat.device)
    need = lib.diinn_metasr_workspace_bytes(b, h, w)
    if workspace is None or workspace.numel() * 4 < need or workspace.device != feat.device:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        stream = torch.cuda.current_stream().cuda_stream
        st = lib.diinn_metasr_decode(C.c_void_p(stream), C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()),
                                     C.c_void_p(workspace.data_ptr()), C.c_void_p(out.data_ptr()), b, h, w, hu, wu)
    _native.check(st, "diinn_metasr_decode")
    return out


# ---------------------------------------------------------------------------
# nn.Module mirror of the reference class
# ---------------------------------------------------------------------------
class ImplicitDecoder(nn.Module):
    """Drop-in for the reference ``ImplicitDecoder`` (diinn.py:39-173).

    Every mode registers the reference's parameters (so any reference checkpoint
    loads); the MI355X kernels implement the paper's final variant, ``mode=3,
    init_q=False`` (README.md:111-112 of the reference), and the ablation modes 1 and 2, whose
    modulation branch depends on the LR cell only.  Mode 4 and ``init_q=True`` raise
    ``NotImplementedError`` in ``forward``."""

    def __init__(self, in_channels: int = 64, hidden_dims=(256, 256, 256, 256), mode: int = 1,
                 init_q: bool = False, sin_mode: int = _native.SIN_DEFAULT, compute: str = "f32"):
        super().__init__()
        self.mode = mode
        self.init_q = init_q
        self.in_channels = in_channels
        self.hidden_dims = list(hidden_dims)
        self.sin_mode = sin_mode
        self.compute = compute            # "f32" (default, reference precision), "bf16", "bf16_full" or "bf16x3"
        unfolded = in_channels * 9
        if init_q:
            self.first_layer = nn.Sequential(nn.Conv2d(3, unfolded, 1), SineAct())
        self.K = nn.ModuleList()
        self.Q = nn.ModuleList()
        k_in, q_in = unfolded, (unfolded if init_q else 3)
        for width in self.hidden_dims:
            self.K.append(nn.Sequential(nn.Conv2d(k_in, width, 1), nn.ReLU()))
            self.Q.append(nn.Sequential(nn.Conv2d(q_in, width, 1), SineAct()))
            # mode 1 chains k -> K[i]; modes 2-4 feed [k or q ; unfolded features] (diinn.py:53-88)
            k_in = width if mode == 1 else width + unfolded
            q_in = width
        if mode == 4:
            self.last_layer = nn.Conv2d(self.hidden_dims[-1], 3, 3, padding=1, padding_mode="reflect")
        else:
            self.last_layer = nn.Conv2d(self.hidden_dims[-1], 3, 1)
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        # P workspaces, one per (device, stream) that called forward: two streams decoding through one module concurrently
        # must not share the hoisted conv's image (at most _MAX_WORKSPACES kept, oldest dropped)
        self._workspaces: "Dict[tuple, torch.Tensor]" = {}

    _MAX_WORKSPACES = 4

    # -- packed-weight cache ---------------------------------------------------
    def _weights_key(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed_weights(self, device) -> torch.Tensor:
        key = self._weights_key(device)
        if self._packed is None or self._packed_key != key:
            sd = self.state_dict()
            self._packed = pack_state_dict(sd, mode=self.mode).to(device)
            self._packed_key = key
        return self._packed

    def _check_supported(self):
        if self.mode not in (1, 2, 3) or self.init_q:
            raise NotImplementedError(
                f"diinn_amd HIP decode path implements modes 1-3 with init_q=False (mode 3 is the reference's "
                f"final model; mode 4's 3x3 head is not tile-independent); got mode={self.mode}, init_q={self.init_q}")
        if self.in_channels != IN_CHANNELS or self.hidden_dims != [HIDDEN] * 4:
            raise NotImplementedError("diinn_amd HIP decode path is built for in_channels=64, hidden_dims=[256]*4")

    def forward(self, x: torch.Tensor, size, bsize: Optional[int] = None) -> torch.Tensor:
        """x [B,64,H,W] fp32 encoder features, size=(H_up, W_up) -> [B,3,H_up,W_up].

        ``bsize`` is the reference's column-strip size (diinn.py:149-160), a
        memory knob there.  The fused kernels keep every per-pixel intermediate
        in registers, so it is accepted and ignored (results are identical for
        any value; the reference's hang for bsize < H_up cannot occur)."""
        self._check_supported()
        _require_cuda(x, "x")
        if bsize is None and torch.is_grad_enabled() and (
                x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # reference: bsize=None runs step() under autograd (training, sr_module.py:128)
            if self.mode != 3 or self.compute != "f32":
                raise NotImplementedError(
                    "diinn_amd: autograd through the HIP decode path covers mode 3 in fp32 (the reference's final "
                    "model); call modes 1/2 or the bf16 path under torch.no_grad()")
            from .training import decode_with_grad
            return decode_with_grad(self, x, size)
        b, c, h, w = x.shape
        need = b * h * w * P_CHANNELS
        if torch.cuda.is_current_stream_capturing():
            # hipGraph capture (modules._GraphReplay): the captured kernels keep the workspace pointer for the
            # graph's lifetime, so it must come from the graph's private pool -- the cached workspace below is
            # replaced (and its block recycled) as soon as a larger input arrives
            workspace = None
        else:
            key = (str(x.device), torch.cuda.current_stream(x.device).cuda_stream)
            workspace = self._workspaces.get(key)
            if workspace is None or workspace.numel() < need:
                self._workspaces.pop(key, None)
                while len(self._workspaces) >= self._MAX_WORKSPACES:
                    self._workspaces.pop(next(iter(self._workspaces)))
                workspace = self._workspaces[key] = torch.empty(need, dtype=torch.float32, device=x.device)
        with torch.no_grad():
            packed = self.packed_weights(x.device)
            return decode_features(x, packed, size, workspace=workspace, sin_mode=self.sin_mode,
                                   compute=self.compute, mode=self.mode)
