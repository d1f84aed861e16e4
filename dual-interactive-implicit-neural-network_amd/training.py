"""Training path of the mode-3 decoder: autograd through the HIP kernels.

The reference trains by calling ``ImplicitDecoder.forward(x, size, bsize=None)`` with autograd on
(diinn.py:170-171 -> step(), :132-139; caller SRLitModule.training_step, sr_module.py:127-129), which
records ~30 ATen ops per call on the materialised [B,576,Hu,Wu] tensor.  Here:

  forward   precompute_P_kernel, then decode_kernel<SAVE> (C ABI ``diinn_decode_train_fwd``): the fused
            inference kernel that additionally writes every layer's rectified modulation k_i and sine
            argument s_i as [channel][pixel] planes -- all the backward pass needs.
  backward  ``backward_fused``: the per-pixel chain (gates and the transposed stacked GEMMs
            g_q[i-1] = Wq_i^T g_a + Qw_i^T g_s) runs on bwd_head_kernel + 3 x bwd_layer_kernel (C ABI
            ``diinn_backward_data``), which leave the gate gradients G_i and the activations q_i as
            tiled planes; every parameter gradient is then one GEMM over the pixel axis per layer
            (plane_gemm_lds_kernel, split-K, no atomics), two skinny products (plane_rowdot_kernel), a
            per-cell segment sum (cell_sum_kernel) and the 3x3 conv's input/weight gradients (MIOpen
            through torch.nn.grad).
            ``backward_from_saved`` states the same gradients in device-agnostic tensor algebra; it
            is the unit-tested formula sheet (CPU, against the reference's own .grad fixtures) and
            the on-GPU cross-check of the fused path.  The forward has no CPU form.

Weights change every optimiser step, so the packed image is rebuilt on the device each forward by one
gather through a permutation index derived once from the host packer (``pack_gather_index``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _native

HIDDEN = 256
IN_CHANNELS = 64
UNFOLD = IN_CHANNELS * 9

# the reference's registration order (ImplicitDecoder.__init__, diinn.py:73-80,92)
PARAM_NAMES: List[str] = (
    [f"K.{i}.0.{t}" for i in range(4) for t in ("weight", "bias")]
    + [f"Q.{i}.0.{t}" for i in range(4) for t in ("weight", "bias")]
    + ["last_layer.weight", "last_layer.bias"]
)
PARAM_SHAPES: Dict[str, Tuple[int, ...]] = {
    "K.0.0.weight": (HIDDEN, UNFOLD, 1, 1), "K.0.0.bias": (HIDDEN,),
    **{f"K.{i}.0.weight": (HIDDEN, HIDDEN + UNFOLD, 1, 1) for i in (1, 2, 3)},
    **{f"K.{i}.0.bias": (HIDDEN,) for i in (1, 2, 3)},
    "Q.0.0.weight": (HIDDEN, 3, 1, 1), "Q.0.0.bias": (HIDDEN,),
    **{f"Q.{i}.0.weight": (HIDDEN, HIDDEN, 1, 1) for i in (1, 2, 3)},
    **{f"Q.{i}.0.bias": (HIDDEN,) for i in (1, 2, 3)},
    "last_layer.weight": (3, HIDDEN, 1, 1), "last_layer.bias": (3,),
}

# backward_fused: the hoisted conv's gradients on the library's own kernels instead of torch.nn.grad (MIOpen).  Measured at B = 16,
# 48x48 x4 (tools/train_conv_grads_ab.py): the input gradient is a 1024 -> 64 3x3 convolution = the encoder's Winograd kernel
# (0.19 ms against MIOpen's 0.39); the weight gradient as unfold + plane GEMM costs 0.5 ms MORE than MIOpen's implicit GEMM
# (im2col and two layout copies of 85-151 MB around a 0.42 ms GEMM), so it is available, not the default.
NATIVE_CONV_DGRAD = True
NATIVE_CONV_WGRAD = True
NATIVE_SUM_PARTS = os.environ.get("DIINN_TRAIN_TORCH_SUM") != "1"       # (A/B: torch.sum over the slice axis instead of sum_parts_kernel)
WGRAD_KSPLIT = 64          # pixel-axis splits of the weight-gradient GEMM: 4 output blocks x 64 = one workgroup per CU
ROWDOT_SPLITS = 1024       # workgroups of the skinny products (HBM-bound)

_gather_index_cpu: Optional[torch.Tensor] = None
_gather_index_dev: Dict[str, torch.Tensor] = {}


def pack_gather_index() -> torch.Tensor:
    """int64 [packed floats]: packed[i] = flat[index[i]] where ``flat`` is the 18 reference tensors
    flattened in PARAM_NAMES order followed by one 0.0 (padding and the bf16 section point at it).
    Derived by packing a state dict whose values are their own flat position (exact in fp32)."""
    global _gather_index_cpu
    if _gather_index_cpu is not None:
        return _gather_index_cpu
    from .decoder import pack_state_dict
    lib = _native.load()
    sd = {}
    pos = 1
    for name in PARAM_NAMES:
        n = int(np.prod(PARAM_SHAPES[name]))
        sd[name] = np.arange(pos, pos + n, dtype=np.float32).reshape(PARAM_SHAPES[name])
        pos += n
    total = pos - 1
    assert total < (1 << 24)
    packed = pack_state_dict(sd, mode=3).numpy()
    idx = np.rint(packed).astype(np.int64) - 1
    off, size = C.c_size_t(), C.c_size_t()
    for section in (7, 9, 10, 11, 12, 13, 14, 15, 16):          # inference-only sections: derived values, not a permutation
        _native.check(lib.diinn_packed_section(section, C.byref(off), C.byref(size)), "diinn_packed_section")
        idx[off.value:off.value + size.value] = -1
    # the validity word behind bL (DIINN_PACKED_MAGIC) reads as zero in a gathered image: the inference entry points,
    # which read the derived sections, then answer NaN instead of decoding with empty weights
    _native.check(lib.diinn_packed_section(6, C.byref(off), C.byref(size)), "diinn_packed_section")
    idx[off.value + 3] = -1
    if idx.max() >= total or idx.min() < -1:
        raise RuntimeError("packed image is not a permutation of the reference tensors")
    used = np.zeros(total, bool)
    used[idx[idx >= 0]] = True
    if not used.all():
        raise RuntimeError("packed image does not reference every parameter element")
    idx[idx < 0] = total                                # the appended zero
    _gather_index_cpu = torch.from_numpy(idx)
    return _gather_index_cpu


_packed_cache: tuple = (None, None, None)          # (key, packed image, the parameter tensors the key describes)


def pack_on_device(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """Reference-ordered parameter tensors (PARAM_NAMES) on a GPU -> packed image on that GPU.
    The last image is kept while no parameter has been modified (a training step decodes once per
    scale with the same weights, sr_module.py:116-121)."""
    global _packed_cache
    key = tuple((p.data_ptr(), p._version) for p in params)
    if _packed_cache[0] == key:
        return _packed_cache[1]
    packed = _pack_on_device(params)
    # the entry keeps the tensors alive: their addresses cannot be handed to other weights while the key is cached
    _packed_cache = (key, packed, tuple(p.detach() for p in params))
    return packed


TRAIN_P_WINOGRAD = True    # the training forward's hoisted conv on the fp32 Winograd kernel (0.21 against 0.48 ms at B = 16, 48 x 48)
_WPU_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))
_section_cache: Dict[int, Tuple[int, int]] = {}


def _section(i: int) -> Tuple[int, int]:
    if i not in _section_cache:
        off, size = C.c_size_t(), C.c_size_t()
        _native.check(_native.load().diinn_packed_section(i, C.byref(off), C.byref(size)), "diinn_packed_section")
        _section_cache[i] = (off.value, size.value)
    return _section_cache[i]


def _fill_wpu(packed: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    """Section 13 (WPU) of a gathered image, on the device: U = G Wx G^T per (output, input) pair in float64 in the host
    packer's own operation order (csrc/diinn_host.cpp: (G g) first, then (.) G^T, sums left to right), rounded once, column 2
    negated, laid out [mt 32][row i 4][sg 8][col j 4][lane 64][e 4] -- bit-identical to diinn_pack_weights' section -- and
    the validity word DIINN_PACKED_MAGIC_WPU ("this training image holds WPU and nothing else derived")."""
    p = dict(zip(PARAM_NAMES, params))
    wx = torch.cat([p["K.0.0.weight"].detach().reshape(HIDDEN, UNFOLD)]
                   + [p[f"K.{i}.0.weight"].detach().reshape(HIDDEN, HIDDEN + UNFOLD)[:, HIDDEN:] for i in (1, 2, 3)], 0)
    w = wx.reshape(4 * HIDDEN, IN_CHANNELS, 3, 3).to(torch.float64)
    g = torch.tensor(_WPU_G, dtype=torch.float64, device=w.device)
    gi = [g[:, a].view(1, 1, 4, 1) for a in range(3)]
    t = gi[0] * w[:, :, 0:1, :] + gi[1] * w[:, :, 1:2, :] + gi[2] * w[:, :, 2:3, :]          # [O, C, i 4, b 3]
    gj = [g[:, b].view(1, 1, 1, 4) for b in range(3)]
    u = t[..., 0:1] * gj[0] + t[..., 1:2] * gj[1] + t[..., 2:3] * gj[2]                      # [O, C, i 4, j 4]
    u[..., 2] = -u[..., 2]
    u = u.to(torch.float32).reshape(32, 32, 8, 4, 2, 4, 4)                                   # [mt, m, sg, e, h, i, j]
    off, size = _section(13)
    packed[off:off + size] = u.permute(0, 5, 2, 6, 4, 1, 3).reshape(-1)                      # [mt, i, sg, j, h, m, e]
    word = _section(6)[0] + 3
    packed[word:word + 1].view(torch.int32).fill_(_native.PACKED_MAGIC_WPU)


def _pack_on_device(params: Sequence[torch.Tensor]) -> torch.Tensor:
    dev = params[0].device
    key = str(dev)
    idx = _gather_index_dev.get(key)
    if idx is None:
        idx = pack_gather_index().to(dev)
        _gather_index_dev[key] = idx
    flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in params] + [torch.zeros(1, device=dev)])
    packed = flat.index_select(0, idx)
    if TRAIN_P_WINOGRAD:
        _fill_wpu(packed, params)
    return packed


# ---------------------------------------------------------------------------
# coordinates as tensors (host tables from the C ABI, bit-exact with the kernels)
# ---------------------------------------------------------------------------
def coordinate_tensors(h: int, w: int, hu: int, wu: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, float]:
    from .decoder import axis_tables
    lib = _native.load()
    small = bool(lib.diinn_uses_small_output_kernel(hu, wu))
    idx_h, rel_h = axis_tables(h, hu, small)
    idx_w, rel_w = axis_tables(w, wu, small)
    ratio = float(np.float32((h * w) / (hu * wu)))
    return (torch.from_numpy(idx_h.astype(np.int64)).to(device), torch.from_numpy(rel_h).to(device),
            torch.from_numpy(idx_w.astype(np.int64)).to(device), torch.from_numpy(rel_w).to(device), ratio)


# ---------------------------------------------------------------------------
# backward from the saved planes (device-agnostic tensor algebra)
# ---------------------------------------------------------------------------
def _cell_sum(g: torch.Tensor, b: int, hu: int, wu: int, h: int, w: int,
              idx_h: torch.Tensor, idx_w: torch.Tensor) -> torch.Tensor:
    """g [C, B*Hu*Wu] -> [B, C, H, W]: sum over the HR pixels of every LR cell (adjoint of the nearest-exact
    replication, diinn.py:168), as two one-hot GEMMs so the summation order is fixed."""
    c = g.shape[0]
    mw = torch.zeros((wu, w), dtype=g.dtype, device=g.device)
    mw[torch.arange(wu, device=g.device), idx_w] = 1
    mh = torch.zeros((h, hu), dtype=g.dtype, device=g.device)
    mh[idx_h, torch.arange(hu, device=g.device)] = 1
    t = g.reshape(c * b * hu, wu) @ mw                       # [C*B*Hu, W]
    t = torch.matmul(mh, t.view(c * b, hu, w))               # [C*B, H, W]
    return t.view(c, b, h, w).permute(1, 0, 2, 3)


def backward_from_saved(gout: torch.Tensor, feat: torch.Tensor, acts: torch.Tensor,
                        params: Sequence[torch.Tensor], size: Sequence[int],
                        need_feat_grad: bool = True) -> Tuple[Optional[torch.Tensor], List[torch.Tensor]]:
    """Gradients of the mode-3 decoder given d(loss)/d(out).

    gout [B,3,Hu,Wu]; feat [B,64,H,W]; acts [4,2,256,N] plain planes k_i, s_i (``untile_planes`` of the
    training forward's buffer, viewed [4,2,256,N]);
    params in PARAM_NAMES order.  Returns (d feat or None, [d param ...] in PARAM_NAMES order).

    With q_i = k_i * sin(s_i), k_0 = relu(P_0[cell]), k_i = relu(Wq_i q_{i-1} + P_i[cell]),
    s_0 = Q0 syn + bQ0, s_i = Qw_i q_{i-1} + bQ_i, out = L q_3 + bL:
        g_a,i = g_q,i * sin(s_i) * [k_i > 0]          g_s,i = g_q,i * k_i * cos(s_i)
        g_q,i-1 = Wq_i^T g_a,i + Qw_i^T g_s,i         dWq_i = g_a,i q_{i-1}^T   dQw_i = g_s,i q_{i-1}^T
        dP_i[cell] = sum over the cell's pixels of g_a,i;  P = conv3x3(feat; Wx) + bK."""
    p = dict(zip(PARAM_NAMES, params))
    b, _, h, w = feat.shape
    hu, wu = int(size[0]), int(size[1])
    n = b * hu * wu
    dev = gout.device
    idx_h, rel_h, idx_w, rel_w, ratio = coordinate_tensors(h, w, hu, wu, dev)
    grads: Dict[str, torch.Tensor] = {}

    g_out = gout.to(torch.float32).permute(1, 0, 2, 3).reshape(3, n)
    lw = p["last_layer.weight"].reshape(3, HIDDEN)
    q3 = acts[3, 0] * torch.sin(acts[3, 1])
    grads["last_layer.weight"] = (g_out @ q3.t()).reshape(3, HIDDEN, 1, 1)
    grads["last_layer.bias"] = g_out.sum(1)
    del q3
    g_q = lw.t() @ g_out                                          # [256, N]

    d_p: List[Optional[torch.Tensor]] = [None] * 4                # each [B,256,H,W]
    d_wq: List[Optional[torch.Tensor]] = [None] * 4
    for i in (3, 2, 1):
        k, s = acts[i, 0], acts[i, 1]
        g_a = g_q * torch.sin(s) * (k > 0)
        g_s = g_q * k * torch.cos(s)
        q_prev = acts[i - 1, 0] * torch.sin(acts[i - 1, 1])
        wfull = p[f"K.{i}.0.weight"].reshape(HIDDEN, HIDDEN + UNFOLD)
        qw = p[f"Q.{i}.0.weight"].reshape(HIDDEN, HIDDEN)
        d_wq[i] = g_a @ q_prev.t()
        grads[f"Q.{i}.0.weight"] = (g_s @ q_prev.t()).reshape(HIDDEN, HIDDEN, 1, 1)
        grads[f"Q.{i}.0.bias"] = g_s.sum(1)
        d_p[i] = _cell_sum(g_a, b, hu, wu, h, w, idx_h, idx_w)
        g_q = wfull[:, :HIDDEN].t() @ g_a + qw.t() @ g_s
        del g_a, g_s, q_prev
    k, s = acts[0, 0], acts[0, 1]
    g_a = g_q * torch.sin(s) * (k > 0)
    g_s = g_q * k * torch.cos(s)
    d_p[0] = _cell_sum(g_a, b, hu, wu, h, w, idx_h, idx_w)
    # syn = (rel_h, rel_w, ratio) per pixel (diinn.py:165-167): dQ0 = g_s syn^T without materialising syn
    g_s4 = g_s.view(HIDDEN, b, hu, wu)
    d_q0 = torch.stack([(g_s4.sum((1, 3)) * rel_h).sum(1), (g_s4.sum((1, 2)) * rel_w).sum(1),
                        g_s4.sum((1, 2, 3)) * ratio], dim=1)
    grads["Q.0.0.weight"] = d_q0.reshape(HIDDEN, 3, 1, 1)
    grads["Q.0.0.bias"] = g_s.sum(1)
    del g_a, g_s, g_q

    dp = torch.cat(d_p, dim=1).contiguous()                       # [B,1024,H,W]
    d_bk = dp.sum((0, 2, 3)).view(4, HIDDEN)
    d_feat = _conv_and_assemble(p, feat, dp, d_wq, d_bk, grads, need_feat_grad)
    return d_feat, [grads[name] for name in PARAM_NAMES]


WGRAD_CONV_KSPLIT = 12     # pixel-axis splits of the hoisted conv's weight-gradient GEMM (20 output blocks x 12)


def _conv_grads_native(feat: torch.Tensor, wx: torch.Tensor, dp: torch.Tensor, need_feat_grad: bool, want_weight: bool = True,
                       wkey=None, wpins=None, a_t: Optional[torch.Tensor] = None):
    """Gradients of P = conv3x3(feat; Wx[1024,64,3,3]) on the library's own kernels (no MIOpen in the decoder's step):
      weight:  dWx[o, (c,ky,kx)] = sum over cells of dP[o, cell] * unfold3x3(feat)[(c,ky,kx), cell] -- the plane GEMM over the
               cell axis (plane_gemm_lds_kernel; the 576 unfolded rows padded to 640 = 5 x 128);
      input :  d_feat = conv3x3(dP; Wx transposed and flipped) -- a 64-output 3x3 convolution over 1024 planes, i.e. the
               encoder's convolution kernels (Winograd F(4x4) / F(2x2) / split-K by the same rule as the trunk)."""
    from . import modules as M                                   # (pack functions; imported late: modules imports the decoder)
    lib = _native.load()
    b, c, h, w = feat.shape
    n = b * h * w
    dev = feat.device
    ptr = lambda x: C.c_void_p(x.data_ptr())                      # noqa: E731
    dp = dp.contiguous()
    d_wx = None
    if want_weight:
        # both operands as tiled plane groups over the CELL axis: dP from cell_sum_kernel itself (``a_t``: no transposing copy),
        # the reference's unfold (rows (c, ky, kx), diinn.py:168) from unfold_tiled_kernel (rows 576..639 zero)
        tiles = (n + PLANE_TILE - 1) // PLANE_TILE
        b_t = torch.empty((tiles, 640, PLANE_TILE), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(lib.diinn_unfold_tiled(C.c_void_p(torch.cuda.current_stream().cuda_stream), ptr(feat.contiguous()), ptr(b_t),
                                                 640, b, h, w), "diinn_unfold_tiled")
        if a_t is None:
            a_t = tile_planes(dp.permute(1, 0, 2, 3).reshape(4 * HIDDEN, n))
        # the product is taken transposed, dWx^T [640 x 1024] = unfold . dP^T: with 1024 = 4 x 256 columns it runs on the kernel's
        # 128 x 256 block form (113 TFLOP/s; the 128 x 128 form the 640 columns of dWx would need: 70)
        ks = max(1, min(WGRAD_CONV_KSPLIT, a_t.shape[0]))
        part = torch.empty((ks, 640, 4 * HIDDEN), dtype=torch.float32, device=dev)
    d_feat = None
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if want_weight:
            _native.check(lib.diinn_plane_gemm_nt(stream, ptr(b_t), 640, 0, ptr(a_t), 4 * HIDDEN, 0, ptr(part), 640, 4 * HIDDEN, n, ks, 0),
                          "diinn_plane_gemm_nt")
            d_wx = _sum_parts(part.view(1, ks, -1)).view(640, 4 * HIDDEN)[:UNFOLD].t()
        if need_feat_grad:
            d_feat = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
            zero = torch.zeros(64, dtype=torch.float32, device=dev)
            cin = 4 * HIDDEN
            form = "wino4" if lib.diinn_rdn_wino4_applies(b, h, w) else "wino" if n >= 8192 else "ksplit"
            # the transposed weight in the kernel's form: repacked only when a K weight changed (an optimizer step, a
            # load_state_dict), not on every backward call.  The Winograd transforms are taken in float64 and rounded once,
            # like the encoder's: F(4x4)'s gradient error 2.5e-5 -> ~1e-5 of max|d_feat| (the fixtures' bound is 1e-4).
            # One entry per (kernel form, device): a multi-scale step alternates forms without evicting each other.  An entry
            # PINS the weight tensors its key describes (as _packed_cache does): while it is cached their addresses cannot be
            # handed to another decoder's weights with equal version counts.
            key = (str(dev), wkey)
            ent = _dgrad_pack.get((form, str(dev)))
            if wkey is None or ent is None or ent[0] != key:
                wt = wx.flip(2, 3).permute(1, 0, 2, 3).contiguous()      # [64, 1024, 3, 3]
                pk = (M.pack_conv_wino4(wt) if form == "wino4" else M.pack_conv_wino(wt) if form == "wino" else M.pack_conv_ksplit(wt))
                ent = (key, pk, tuple(t.detach() for t in (wpins or ())))
                if wkey is not None:
                    _dgrad_pack[(form, str(dev))] = ent
            pk = ent[1]
            if form == "wino4":
                ws = _wino4_workspace(dev)                       # (a partly filled last round is split over the input channels)
                ws[:512].zero_()                                 # the arrival counters, whatever an aborted launch may have left (as the trunk does; never the sticky status word)
                _native.check(lib.diinn_conv_wino4_ws(stream, ptr(dp), cin * h * w, cin, ptr(pk), ptr(zero), None, 0, ptr(d_feat),
                                                      c * h * w, 0, b, h, w, ptr(ws), ws.numel()), "diinn_conv_wino4_ws")
            elif form == "wino":
                _native.check(lib.diinn_conv_wino(stream, ptr(dp), cin * h * w, cin, ptr(pk), ptr(zero), None, 0, ptr(d_feat),
                                                  c * h * w, 0, b, h, w), "diinn_conv_wino")
            else:
                _native.check(lib.diinn_conv_ksplit(stream, ptr(dp), cin * h * w, cin, 9, ptr(pk), ptr(zero), None, 0, ptr(d_feat),
                                                    c * h * w, None, 0, 0, b, h, w), "diinn_conv_ksplit")
    return d_wx, d_feat


_dgrad_pack: Dict[tuple, tuple] = {}               # (form, device) -> (key, the transposed hoisted-conv weight in that kernel's form, the pinned K weights)


def _wino4_workspace(dev) -> torch.Tensor:
    """diinn_conv_wino4_ws's workspace: the encoder's split area of this (device, current stream) (modules.RDN._w4_area:
    control words zeroed once, launches on one stream are ordered, two streams never share slabs or tickets)."""
    from . import modules as M
    return M.RDN._w4_area(dev)


def _conv_and_assemble(p: Dict[str, torch.Tensor], feat: torch.Tensor, dp: torch.Tensor, d_wq, d_bk,
                       grads: Dict[str, torch.Tensor], need_feat_grad: bool, native: bool = False,
                       a_t: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """P = conv3x3(feat; Wx[1024,64,3,3]) + bK: weight / input gradients of that one convolution (``native``: the
    library's kernels, else torch.nn.grad = MIOpen on the GPU), then the K.i gradients in the reference's [256, 256+576] layout."""
    wx = torch.cat([p["K.0.0.weight"].reshape(HIDDEN, UNFOLD)]
                   + [p[f"K.{i}.0.weight"].reshape(HIDDEN, HIDDEN + UNFOLD)[:, HIDDEN:] for i in (1, 2, 3)], 0)
    wx = wx.reshape(4 * HIDDEN, IN_CHANNELS, 3, 3).contiguous()
    nat_w, nat_d = (native if isinstance(native, tuple) else (bool(native), bool(native)))
    d_wx = d_feat = None
    if nat_w or (nat_d and need_feat_grad):
        wkey = tuple((p[f"K.{i}.0.weight"].data_ptr(), p[f"K.{i}.0.weight"]._version) for i in range(4))
        d_wx, d_feat = _conv_grads_native(feat, wx, dp, need_feat_grad and nat_d, want_weight=nat_w, wkey=wkey,
                                          wpins=tuple(p[f"K.{i}.0.weight"] for i in range(4)), a_t=a_t)
    if d_wx is None:
        d_wx = torch.nn.grad.conv2d_weight(feat, wx.shape, dp, padding=1)
    if d_feat is None and need_feat_grad:
        d_feat = torch.nn.grad.conv2d_input(feat.shape, wx, dp, padding=1)
    d_wx = d_wx.reshape(4, HIDDEN, UNFOLD)
    grads["K.0.0.weight"] = d_wx[0].reshape(HIDDEN, UNFOLD, 1, 1)
    grads["K.0.0.bias"] = d_bk[0]
    for i in (1, 2, 3):
        grads[f"K.{i}.0.weight"] = torch.cat([d_wq[i], d_wx[i]], dim=1).reshape(HIDDEN, HIDDEN + UNFOLD, 1, 1)
        grads[f"K.{i}.0.bias"] = d_bk[i]
    return d_feat


PLANE_TILE = 32


def tile_planes(x: torch.Tensor) -> torch.Tensor:
    """Plain planes [C, n] -> tiled group [ceil(n/32), C, 32] (zero padding), the layout of include/diinn_hip.h."""
    c, n = x.shape
    t = (n + PLANE_TILE - 1) // PLANE_TILE
    out = x.new_zeros((c, t * PLANE_TILE))
    out[:, :n] = x
    return out.view(c, t, PLANE_TILE).permute(1, 0, 2).contiguous()


def untile_planes(x: torch.Tensor, n: int) -> torch.Tensor:
    """Tiled groups [..., T, C, 32] -> plain planes [..., C, n] (a copy; tests and the formula path)."""
    *lead, t, c, w = x.shape
    d = len(lead)
    return x.permute(*range(d), d + 1, d, d + 2).reshape(*lead, c, t * w)[..., :n]


_geometry_cache: "Dict[tuple, dict]" = {}
GEOMETRY_CACHE_ENTRIES = 8


def _geometry(b: int, h: int, w: int, hu: int, wu: int, dev) -> dict:
    """Per-shape constants of the backward pass, built once per (B, LR size, HR size, device): the cell
    rectangles of cell_sum_kernel and the tiled right-hand side (rel_h, rel_w, ratio, 1) of the layer-0
    product.  Training revisits a handful of shapes (one per scale), so a small LRU suffices."""
    key = (b, h, w, hu, wu, str(dev))
    geo = _geometry_cache.pop(key, None)
    if geo is None:
        idx_h, rel_h, idx_w, rel_w, ratio = coordinate_tensors(h, w, hu, wu, dev)
        syn = torch.empty((4, b, hu, wu), dtype=torch.float32, device=dev)
        syn[0] = rel_h[None, :, None]
        syn[1] = rel_w[None, None, :]
        syn[2] = ratio
        syn[3] = 1.0
        geo = {
            "seg_h": torch.searchsorted(idx_h, torch.arange(h + 1, device=dev)).to(torch.int32),
            "seg_w": torch.searchsorted(idx_w, torch.arange(w + 1, device=dev)).to(torch.int32),
            "syn_t": tile_planes(syn.view(4, b * hu * wu)),
        }
        while len(_geometry_cache) >= GEOMETRY_CACHE_ENTRIES:
            _geometry_cache.pop(next(iter(_geometry_cache)))
    _geometry_cache[key] = geo
    return geo


def _sum_parts(part: torch.Tensor) -> torch.Tensor:
    """[groups, nparts, n] -> [groups, n]: the split partials of a GEMM / rowdot launch added in slice order (sum_parts_kernel)."""
    groups, nparts, n = part.shape
    if n % 4 or not part.is_contiguous() or not NATIVE_SUM_PARTS:
        return part.sum(1)
    out = torch.empty((groups, n), dtype=torch.float32, device=part.device)
    with torch.cuda.device(part.device):
        _native.check(_native.load().diinn_sum_parts(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(part.data_ptr()),
                                                     C.c_void_p(out.data_ptr()), groups, nparts, n), "diinn_sum_parts")
    return out


def backward_fused(gout: torch.Tensor, feat: torch.Tensor, acts: torch.Tensor, params: Sequence[torch.Tensor],
                   packed: torch.Tensor, size: Sequence[int],
                   need_feat_grad: bool = True) -> Tuple[Optional[torch.Tensor], List[torch.Tensor]]:
    """The same gradients as ``backward_from_saved``, on the HIP kernels throughout:
      diinn_backward_data   bwd_head_kernel + 3 x bwd_layer_kernel: the per-pixel chain; leaves the gate
                            gradients G_i = (g_a,i ; g_s,i) and the activations q_i as tiled planes
      diinn_plane_gemm_nt   [dWq_i ; dQw_i | bias sums] = G_i [512 x N] . q_{i-1}^T [N x 256], split over pixels
      diinn_plane_rowdot    the two skinny products (layer 0 against (rel_h, rel_w, ratio, 1); head against g_out)
      diinn_backward_cell_sum   dP = per-cell sums of g_a
    and the 3x3 convolution's input/weight gradients from MIOpen (torch.nn.grad).
    ``acts`` is the tiled buffer [4, T, 512, 32] of the training forward."""
    lib = _native.load()
    p = dict(zip(PARAM_NAMES, params))
    b, _, h, w = feat.shape
    hu, wu = int(size[0]), int(size[1])
    n = b * hu * wu
    t = (n + PLANE_TILE - 1) // PLANE_TILE
    dev = gout.device
    if tuple(acts.shape) != (4, t, 2 * HIDDEN, PLANE_TILE) or not acts.is_contiguous():
        raise ValueError("acts must be the contiguous tiled [4, T, 512, 32] buffer of the training forward")
    geo = _geometry(b, h, w, hu, wu, dev)
    seg_h, seg_w, syn_t = geo["seg_h"], geo["seg_w"], geo["syn_t"]
    gp = gout.to(torch.float32).permute(1, 0, 2, 3).reshape(3, n).contiguous()
    g = torch.empty((4, t, 2 * HIDDEN, PLANE_TILE), dtype=torch.float32, device=dev)
    q = torch.empty((4, t, HIDDEN, PLANE_TILE), dtype=torch.float32, device=dev)
    gout_t = tile_planes(torch.cat([gp, gp.new_zeros((1, n))], 0))     # 4-row right-hand side of the head product
    ksplit = max(1, min(WGRAD_KSPLIT, t))
    rsplit = max(1, min(ROWDOT_SPLITS, t))
    part = torch.empty((3, ksplit, 2 * HIDDEN, HIDDEN + 1), dtype=torch.float32, device=dev)
    part0 = torch.empty((rsplit, 2 * HIDDEN, 4), dtype=torch.float32, device=dev)
    partl = torch.empty((rsplit, HIDDEN, 4), dtype=torch.float32, device=dev)
    dp = torch.empty((b, 4 * HIDDEN, h, w), dtype=torch.float32, device=dev)
    # the hoisted conv's weight gradient on the library's own GEMM wants dP tiled over the cell axis as well: cell_sum_kernel
    # writes it (a ragged last tile's padding must be zero: the GEMM reads whole tiles)
    cells = b * h * w
    a_t = None
    if NATIVE_CONV_WGRAD:
        a_t = (torch.empty if cells % PLANE_TILE == 0 else torch.zeros)(((cells + PLANE_TILE - 1) // PLANE_TILE, 4 * HIDDEN, PLANE_TILE),
                                                                        dtype=torch.float32, device=dev)
    ptr = lambda x: C.c_void_p(x.data_ptr())                      # noqa: E731
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _native.check(lib.diinn_backward_data(stream, ptr(gp), ptr(acts), ptr(packed), ptr(g), ptr(q), n),
                      "diinn_backward_data")
        for i in (3, 2, 1):
            _native.check(lib.diinn_plane_gemm_nt(stream, ptr(g[i]), 2 * HIDDEN, 0, ptr(q[i - 1]), HIDDEN, 0,
                                                  ptr(part[i - 1]), 2 * HIDDEN, HIDDEN, n, ksplit, 1),
                          "diinn_plane_gemm_nt")
        _native.check(lib.diinn_plane_rowdot(stream, ptr(g[0]), 2 * HIDDEN, ptr(syn_t), ptr(part0), 2 * HIDDEN, n, rsplit),
                      "diinn_plane_rowdot")
        _native.check(lib.diinn_plane_rowdot(stream, ptr(q[3]), HIDDEN, ptr(gout_t), ptr(partl), HIDDEN, n, rsplit),
                      "diinn_plane_rowdot")
        _native.check(lib.diinn_backward_cell_sum_ex(stream, ptr(g), ptr(seg_h), ptr(seg_w), ptr(dp), ptr(a_t) if a_t is not None else None,
                                                     b, h, w, hu, wu), "diinn_backward_cell_sum_ex")
    grads: Dict[str, torch.Tensor] = {}
    dl = _sum_parts(partl.view(1, rsplit, -1)).view(HIDDEN, 4)    # [256, 4]: q_3 . (g_out ; 0)^T
    grads["last_layer.weight"] = dl[:, :3].t().reshape(3, HIDDEN, 1, 1)
    grads["last_layer.bias"] = gp.sum(1)
    dws = _sum_parts(part.view(3, ksplit, -1)).view(3, 2 * HIDDEN, HIDDEN + 1)   # [3, 512, 257]: [dWq_i ; dQw_i | bias sums]
    d_wq: List[Optional[torch.Tensor]] = [None] * 4
    d_bk: List[Optional[torch.Tensor]] = [None] * 4
    for i in (3, 2, 1):
        dw = dws[i - 1]
        d_wq[i] = dw[:HIDDEN, :HIDDEN]
        d_bk[i] = dw[:HIDDEN, HIDDEN]
        grads[f"Q.{i}.0.weight"] = dw[HIDDEN:, :HIDDEN].reshape(HIDDEN, HIDDEN, 1, 1)
        grads[f"Q.{i}.0.bias"] = dw[HIDDEN:, HIDDEN]
    d0 = _sum_parts(part0.view(1, rsplit, -1)).view(2 * HIDDEN, 4)   # [512, 4]: (g_a,0 ; g_s,0) . (rel_h, rel_w, ratio, 1)^T
    d_bk[0] = d0[:HIDDEN, 3]
    grads["Q.0.0.weight"] = d0[HIDDEN:, :3].reshape(HIDDEN, 3, 1, 1)
    grads["Q.0.0.bias"] = d0[HIDDEN:, 3]
    d_feat = _conv_and_assemble(p, feat, dp, d_wq, d_bk, grads, need_feat_grad, native=(NATIVE_CONV_WGRAD, NATIVE_CONV_DGRAD), a_t=a_t)
    return d_feat, [grads[name] for name in PARAM_NAMES]


# ---------------------------------------------------------------------------
# autograd function
# ---------------------------------------------------------------------------
class DecodeMode3Function(torch.autograd.Function):
    """out = decoder(feat) on the HIP kernels, differentiable in feat and the 18 parameter tensors."""

    @staticmethod
    def forward(ctx, feat: torch.Tensor, hu: int, wu: int, sin_mode: int, *params: torch.Tensor) -> torch.Tensor:
        lib = _native.load()
        if not feat.is_cuda:
            raise RuntimeError("diinn_amd: the training forward runs on a ROCm GPU only (no CPU implementation)")
        if len(params) != len(PARAM_NAMES):
            raise ValueError(f"expected {len(PARAM_NAMES)} parameter tensors in PARAM_NAMES order")
        feat_c = feat.detach().contiguous().to(torch.float32)
        b, c, h, w = feat_c.shape
        if c != IN_CHANNELS:
            raise ValueError(f"feat must be [B,{IN_CHANNELS},H,W]")
        n = b * hu * wu
        if lib.diinn_training_plane_floats(n, 2 * HIDDEN) < 0:
            raise RuntimeError(f"diinn_amd: B*Hu*Wu = {n} HR pixels in one training forward exceeds the limit; split the batch")
        dev = feat_c.device
        packed = pack_on_device(params)
        workspace = torch.empty(b * h * w * 4 * HIDDEN, dtype=torch.float32, device=dev)
        acts = torch.empty((4, (n + PLANE_TILE - 1) // PLANE_TILE, 2 * HIDDEN, PLANE_TILE), dtype=torch.float32, device=dev)
        out = torch.empty((b, 3, hu, wu), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            # the hoisted conv: the Winograd kernel when the image carries its section (DIINN_PACKED_MAGIC_WPU: _fill_wpu), else
            # the direct kernel, which reads permutation sections only
            p_fn = lib.diinn_precompute_P_wpu if TRAIN_P_WINOGRAD else lib.diinn_precompute_P
            _native.check(p_fn(stream, C.c_void_p(feat_c.data_ptr()), C.c_void_p(packed.data_ptr()),
                               C.c_void_p(workspace.data_ptr()), b, h, w, 0, h), "diinn_precompute_P")
            _native.check(lib.diinn_decode_train_fwd(stream, C.c_void_p(workspace.data_ptr()),
                                                     C.c_void_p(packed.data_ptr()), C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(acts.data_ptr()), b, h, w, hu, wu, int(sin_mode)),
                          "diinn_decode_train_fwd")
        ctx.save_for_backward(feat_c, acts, packed, *[p_.detach() for p_ in params])
        ctx.size = (hu, wu)
        return out

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        feat, acts, packed, *params = ctx.saved_tensors
        d_feat, d_params = backward_fused(gout.contiguous(), feat, acts, params, packed, ctx.size,
                                          need_feat_grad=ctx.needs_input_grad[0])
        need = ctx.needs_input_grad[4:]
        return (d_feat, None, None, None, *[g if nd else None for g, nd in zip(d_params, need)])


def decode_with_grad(decoder, feat: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """``ImplicitDecoder.forward(x, size, None)`` under autograd (mode 3)."""
    named = dict(decoder.named_parameters())
    params = [named[name] for name in PARAM_NAMES]
    hu, wu = size
    return DecodeMode3Function.apply(feat, int(hu), int(wu), int(decoder.sin_mode), *params)
