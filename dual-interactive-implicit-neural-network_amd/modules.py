"""The callers' side of the boundary: the reference's module
surface around the decoder, so its workflows run on the MI355X path unchanged.

  RDN / make_rdn     encoder, plain PyTorch-ROCm (north_star: "the encoder stays PyTorch-ROCm");
                     parameter names follow reference src/models/components/rdn.py:37-105 so
                     checkpoints load
  DIINN              reference diinn.py:8-19: encoder -> ImplicitDecoder (HIP kernels)
  make_net           reference sr_module.py:42-50
  SRLitModule        reference sr_module.py:62-194, the parts the inference callers use:
                     ctor hparams, ``net``, ``sub``/``div`` buffers, forward, step,
                     training_step / validation_step / test_step, configure_optimizers,
                     load_from_checkpoint.  pytorch_lightning is not required (it is absent from
                     the target image); scripts/train.py is the plain loop around these hooks.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .decoder import ImplicitDecoder
from .metrics import calc_psnr, psnr, resize_fn, ssim  # noqa: F401  (re-exported like the reference's sr_module)


# ---------------------------------------------------------------------------
# RDN encoder (config 'B': 16 blocks x 8 convs, growth 64), features only
# ---------------------------------------------------------------------------
class RDB_Conv(nn.Module):
    """3x3 conv + ReLU whose output is concatenated to its input (dense connection)."""

    def __init__(self, inChannels: int, growRate: int, kSize: int = 3):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(inChannels, growRate, kSize, padding=(kSize - 1) // 2), nn.ReLU())

    def forward(self, x):
        return torch.cat((x, self.conv(x)), 1)


class RDB(nn.Module):
    """Residual dense block: C dense convs, 1x1 local feature fusion, residual add."""

    def __init__(self, growRate0: int, growRate: int, nConvLayers: int, kSize: int = 3):
        super().__init__()
        self.convs = nn.Sequential(*[RDB_Conv(growRate0 + c * growRate, growRate, kSize) for c in range(nConvLayers)])
        self.LFF = nn.Conv2d(growRate0 + nConvLayers * growRate, growRate0, 1)

    def forward(self, x):
        if not torch.is_grad_enabled():
            return self._forward_dense_buffer(x)
        return self.LFF(self.convs(x)) + x

    def _forward_dense_buffer(self, x):
        """Inference form of the dense block: one [B, G0 + C*G, H, W] buffer, every conv reads the
        channels written so far and appends its G outputs in place -- the reference's
        ``torch.cat((x, out), 1)`` (rdn.py:15-17) re-copies the whole growing stack at every conv
        (2,816 channel-planes per block instead of 576).  Same arithmetic, same results."""
        b, g0, h, w = x.shape
        convs = list(self.convs)
        g = convs[0].conv[0].out_channels
        buf = x.new_empty((b, g0 + len(convs) * g, h, w))
        buf[:, :g0] = x
        c_in = g0
        for layer in convs:
            conv = layer.conv[0]
            buf[:, c_in:c_in + g] = F.relu(F.conv2d(buf[:, :c_in], conv.weight, conv.bias, padding=conv.padding))
            c_in += g
        return self.LFF(buf) + x


def pack_conv_ksplit(weight: torch.Tensor) -> torch.Tensor:
    """Conv weight [64, Cin, kh, kw] (Cin % 64 == 0; 3x3 or 1x1) -> the layout ``diinn_conv_ksplit`` reads
    (include/diinn_hip.h): [half 2][wave 8][tap][group][lane 64][4] with cout = 32 half + (lane & 31) and
    input channel = wave*Cin/8 + 8 group + 2 e + (lane >> 5)."""
    co, cin, kh, kw = weight.shape
    if co != 64 or cin % 64 or (kh, kw) not in ((3, 3), (1, 1)):
        raise ValueError(f"unsupported convolution shape {tuple(weight.shape)}")
    taps, groups = kh * kw, cin // 64
    w = weight.detach().to(torch.float32).reshape(2, 32, 8, groups, 4, 2, taps)     # [half, i, wave, g, e, h, tap]
    return w.permute(0, 2, 6, 3, 5, 1, 4).reshape(-1)                               # [half, wave, tap, g, h, i, e]


def _winograd_weight(weight: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """U = G W G^T per (output, input) pair, [O, C, 3, 3] -> [O, C, R, R], in ``g``'s dtype as plain broadcast products and sums
    (left to right, (G W) first: the host packer's order for the decoder's hoisted conv) -- not an einsum, which would run a
    library GEMM per call (the training step re-packs the input gradient's weight whenever a parameter changed)."""
    w = weight.detach().to(g.dtype)
    r = g.shape[0]
    gi = [g[:, a].view(1, 1, r, 1) for a in range(3)]
    t = gi[0] * w[:, :, 0:1, :] + gi[1] * w[:, :, 1:2, :] + gi[2] * w[:, :, 2:3, :]          # [O, C, R, 3]
    gj = [g[:, b].view(1, 1, 1, r) for b in range(3)]
    return t[..., 0:1] * gj[0] + t[..., 1:2] * gj[1] + t[..., 2:3] * gj[2]                   # [O, C, R, R]


def pack_conv_wino(weight: torch.Tensor, dtype: torch.dtype = torch.float64) -> torch.Tensor:
    """3x3 conv weight [64, Cin, 3, 3] (Cin % 8 == 0) -> the Winograd F(2x2, 3x3) image ``diinn_conv_wino`` reads
    (include/diinn_hip.h): U = G W G^T per (output, input) pair, computed in float64 and rounded once, laid out
    [row i 4][chunk Cin/8][col j 4][half 2][lane 64][4] with cout = 32 half + (lane & 31) and input channel =
    8 chunk + 2 e + (lane >> 5); column j = 2 is stored negated."""
    co, cin, kh, kw = weight.shape
    if co != 64 or cin % 8 or (kh, kw) != (3, 3):
        raise ValueError(f"unsupported convolution shape {tuple(weight.shape)}")
    g = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=dtype,
                     device=weight.device)                       # on the weight's device: 1.4 s for the trunk on the CPU, ms on the GPU
    u = _winograd_weight(weight, g).to(torch.float32)
    u[..., 2] = -u[..., 2]                                      # the kernel's input transform produces column 2 negated
    u = u.reshape(2, 32, cin // 8, 4, 2, 4, 4)                  # [half, m, chunk, e, h, i, j]
    return u.permute(5, 2, 6, 0, 4, 1, 3).reshape(-1)           # [i, chunk, j, half, h, m, e]


_WINO4_G = ((1 / 4, 0, 0), (-1 / 6, -1 / 6, -1 / 6), (-1 / 6, 1 / 6, -1 / 6), (1 / 24, 1 / 12, 1 / 6), (1 / 24, -1 / 12, 1 / 6), (0, 0, 1))


def pack_conv_wino4(weight: torch.Tensor, dtype: torch.dtype = torch.float64) -> torch.Tensor:
    """3x3 conv weight [64, Cin, 3, 3] (Cin % 8 == 0) -> the Winograd F(4x4, 3x3) image ``diinn_conv_wino4`` reads
    (include/diinn_hip.h): U = G W G^T (6x6 per (output, input) pair), computed in float64 and rounded once, laid out
    [wave 12][half 2][chunk Cin/8][q 3][lane 64][4] with position 6 i + j = 3 wave + q, cout = 32 half + (lane & 31)
    and input channel = 8 chunk + 2 e + (lane >> 5)."""
    co, cin, kh, kw = weight.shape
    if co != 64 or cin % 8 or (kh, kw) != (3, 3):
        raise ValueError(f"unsupported convolution shape {tuple(weight.shape)}")
    g = torch.tensor(_WINO4_G, dtype=dtype, device=weight.device)
    u = _winograd_weight(weight, g).to(torch.float32)
    u = u.reshape(2, 32, cin // 8, 4, 2, 12, 3)                 # [half, m, chunk, e, h, wave, q]
    return u.permute(5, 0, 2, 6, 4, 1, 3).reshape(-1)           # [wave, half, chunk, q, h, m, e]


def pack_conv_x3(weight: torch.Tensor) -> torch.Tensor:
    """Conv weight [64, Cin, k, k] (k = 3 or 1, Cin % 16 == 0) -> the split-bf16 image ``diinn_conv3x3_x3`` and the trunk's
    split-bf16 fusion layer read (include/diinn_hip.h): every weight as hi = bf16(w), lo = bf16(w - hi), laid out
    [group Cin/16][tap k*k][M-tile 2][hi, lo][lane 64][8 bf16] with cout = 32 mt + (lane & 31) and input channel =
    16 group + 8 (lane >> 5) + j; returned as float32 words (two bf16 each), k*k * 64 * Cin of them."""
    co, cin, kh, kw = weight.shape
    if co != 64 or cin % 16 or (kh, kw) not in ((3, 3), (1, 1)):
        raise ValueError(f"unsupported convolution shape {tuple(weight.shape)}")
    taps = kh * kw
    w = weight.detach().to(torch.float32)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    parts = torch.stack([hi, lo], 0).reshape(2, 2, 32, cin // 16, 2, 8, taps)       # [part, mt, m, g, h, j, tap]
    img = parts.permute(3, 6, 1, 0, 4, 2, 5).contiguous()                           # [g, tap, mt, part, h, m, j]
    return img.view(torch.int16).reshape(-1, 2).view(torch.int32).reshape(-1).view(torch.float32)


class RDN(nn.Module):
    _CONFIGS = {"A": (20, 6, 32), "B": (16, 8, 64)}
    # Inference (no autograd, fp32, config 'B') runs the whole encoder on the library's kernels:
    # SFENet1 on diinn_sfe1_forward, the 147 convolutions after it through diinn_rdn_forward_ex -- the split-K
    # kernel on small maps, Winograd 3x3 + streaming 1x1 kernels from 8192 pixels on.  Measured (tools/enc_trunk_time.py,
    # HIP vs MIOpen eager): 1.95 vs 7.6 ms at 48x48, 3.6 vs 7.6 at 96x96, 3.9 vs 8.1 at 128x128, 9.3 vs 20.2 at 192x192,
    # 11.8 vs 28.6 at 256x256, 28.3 vs 61.6 at 384x384, 47.0 vs 109.5 at 512x512.  The attribute caps the batch*H*W
    # that takes this path (workspace: 2,240 floats per pixel + 34.6 MB for the F(4x4) kernel's split); None disables it (MIOpen everywhere).
    hip_trunk_max_pixels: Optional[int] = 1024 * 1024          # byte offsets of a wave's channel slice stay far below 2^31
    # 3x3 layers as Winograd F(2x2, 3x3) (csrc/diinn_winograd.hip) on maps of >= 8192 pixels: 2.25x fewer MFMAs, fp32,
    # equal to the direct sum up to reassociation (~1e-6 relative).  False keeps every layer on the direct kernel.
    hip_winograd: bool = True
    # ... and as Winograd F(4x4, 3x3) (csrc/diinn_winograd4.hip) where that kernel needs fewer rounds of workgroups (from
    # about 190 x 190 pixels on: diinn_rdn_wino4_applies): 1.78x fewer MFMAs again; 256x256 11.8 -> 8.7 ms, 512x512 46.0 ->
    # 34.1 ms.  Per layer ~1e-5 of max|out| on unit-variance inputs (F(2x2): 4e-7); through the whole trunk 2e-6 absolute
    # at max|feat| 1.6 against MIOpen (F(2x2): 6e-7; the parity bound is 2e-5 x max).  False keeps F(2x2, 3x3) everywhere.
    hip_winograd4: bool = True
    # optional: the 3x3 layers in split-bf16 arithmetic on the bf16 MFMA (csrc/diinn_conv_x3.hip) on maps of >= 32,768
    # pixels: hi + lo bf16 operands, three products per term, fp32 accumulation; inside a dense block the activations are
    # exchanged already split.  Per layer ~4e-6 of max|out| against float64; the whole trunk differs from the fp32 one by
    # ~3e-6 of max|feat| and the decoded image by ~3e-8.  Measured per trunk: 192x192 9.4 -> 6.7 ms,
    # 256x256 11.85 -> 7.9 ms, 384x384 28.1 -> 20.1 ms, 512x512 46.6 -> 31.4 ms; below ~180x180 the Winograd kernels stay faster and are used.
    hip_split_bf16: bool = False

    def __init__(self, G0: int = 64, RDNkSize: int = 3, RDNconfig: str = "B", n_colors: int = 3):
        super().__init__()
        self.D, C, G = self._CONFIGS[RDNconfig]
        pad = (RDNkSize - 1) // 2
        self.SFENet1 = nn.Conv2d(n_colors, G0, RDNkSize, padding=pad)
        self.SFENet2 = nn.Conv2d(G0, G0, RDNkSize, padding=pad)
        self.RDBs = nn.ModuleList([RDB(G0, G, C) for _ in range(self.D)])
        self.GFF = nn.Sequential(nn.Conv2d(self.D * G0, G0, 1), nn.Conv2d(G0, G0, RDNkSize, padding=pad))
        self.out_dim = G0
        self.args = SimpleNamespace(G0=G0, RDNkSize=RDNkSize, RDNconfig=RDNconfig, scale=[2],
                                    no_upsampling=True, n_colors=n_colors)
        self._hip_ok = (RDNconfig == "B" and G0 == 64 and RDNkSize == 3)
        self._hip_pack = None
        self._hip_key = None

    def _trunk_layers(self):
        """The 147 convolutions after SFENet1 in execution order (rdn.py:97-103)."""
        layers = [self.SFENet2]
        for rdb in self.RDBs:
            layers += [c.conv[0] for c in rdb.convs] + [rdb.LFF]
        return layers + [self.GFF[0], self.GFF[1]]

    def _hip_packed(self, device):
        layers = self._trunk_layers()
        key = (str(device),) + tuple((l.weight.data_ptr(), l.weight._version, l.bias._version) for l in layers)
        if self._hip_pack is None or self._hip_key != key:
            w = torch.cat([pack_conv_ksplit(l.weight) for l in layers]).to(device)
            b = torch.cat([l.bias.detach().to(torch.float32) for l in layers]).to(device)
            self._hip_pack, self._hip_key, self._hip_x3, self._hip_wu4, self._hip_wu2 = (w, b), key, None, None, None
        return self._hip_pack

    def _hip_packed_wino(self, device):
        """The F(2x2, 3x3) image of the 130 3x3 weights (152 MB), built on first use: a model that only sees maps whose
        3x3 layers run F(4x4, 3x3) (341 MB) or the split-K kernel (the 85 MB permutation above) never builds it.
        ``RDN.free_unused_images()`` drops the images again."""
        self._hip_packed(device)
        if getattr(self, "_hip_wu2", None) is None:
            self._hip_wu2 = torch.cat([pack_conv_wino(l.weight.to(device)) for l in self._trunk_layers()
                                       if l.kernel_size == (3, 3)]).to(device)
        return self._hip_wu2

    def free_unused_images(self, keep=()):
        """Drop the derived weight images (``"wino"`` F(2x2): 152 MB, ``"wino4"`` F(4x4): 341 MB, ``"x3"`` split bf16: 88 MB)
        that are not named in ``keep``; they are rebuilt on the next forward that needs them.  Images captured by a hipGraph
        entry stay alive with it."""
        for name, attr in (("wino", "_hip_wu2"), ("wino4", "_hip_wu4"), ("x3", "_hip_x3")):
            if name not in keep:
                setattr(self, attr, None)

    def _hip_packed_wino4(self, device):
        """The F(4x4, 3x3) image of the 130 3x3 weights, built on first use (and again when a weight changes)."""
        self._hip_packed(device)
        if getattr(self, "_hip_wu4", None) is None:
            self._hip_wu4 = torch.cat([pack_conv_wino4(l.weight.to(device)) for l in self._trunk_layers()
                                       if l.kernel_size == (3, 3)]).to(device)
        return self._hip_wu4

    def _hip_packed_x3(self, device):
        """The split-bf16 image of the 130 3x3 weights, built on first use (and again when a weight changes)."""
        self._hip_packed(device)
        if getattr(self, "_hip_x3", None) is None:
            layers = self._trunk_layers()
            # the 130 3x3 layers in execution order, then the 16 local-fusion 1x1 layers
            self._hip_x3 = torch.cat([pack_conv_x3(l.weight.to(device)) for l in layers if l.kernel_size == (3, 3)] +
                                     [pack_conv_x3(rdb.LFF.weight.to(device)) for rdb in self.RDBs]).to(device)
        return self._hip_x3

    # the F(4x4) kernel's split area (34.6 MB: control words + partial-output slabs), ONE per (device, stream), kept across
    # forwards: its sticky status word then remembers a hand-off that ever gave up (handoff_status()); two streams never
    # share slabs or tickets.  Class-wide: every RDN of the process on that (device, stream) may use it (launches on one
    # stream are ordered).
    _w4_areas: dict = {}
    _W4_AREAS_MAX = 16

    @classmethod
    def _w4_area(cls, device):
        from . import _native
        floats = _native.load().diinn_conv_wino4_workspace_floats()
        if torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture the area comes from the graph's private pool and lives with the graph (its control
            # words are zeroed by a captured memset: a replay starts clean; a give-up is still NaN in that replay's output)
            ws = torch.empty(floats, dtype=torch.float32, device=device)
            ws[:1024].zero_()
            return ws
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = cls._w4_areas.get(key)
        if ws is None:
            while len(cls._w4_areas) >= cls._W4_AREAS_MAX:       # streams come and go: the oldest area goes with them
                cls._w4_areas.pop(next(iter(cls._w4_areas)))
            ws = torch.empty(floats, dtype=torch.float32, device=device)
            ws[:1024].zero_()                                 # the control words, once (a forward re-zeroes only the counters)
            cls._w4_areas[key] = ws
        return ws

    @classmethod
    def handoff_status(cls, clear: bool = True) -> int:
        """1 if the F(4x4) kernel's cross-workgroup hand-off has ever given up on one of this process' split areas (the
        features computed then, and since, are NaN: the failure is loud on the device already); synchronises the streams
        concerned.  ``clear`` re-arms the areas."""
        import ctypes as C
        from . import _native
        lib = _native.load()
        worst = 0
        for (dev, stream), ws in list(cls._w4_areas.items()):
            st = C.c_int(0)
            with torch.cuda.device(ws.device):
                _native.check(lib.diinn_conv_wino4_ws_status(C.c_void_p(stream), C.c_void_p(ws.data_ptr()), int(clear), C.byref(st)),
                              "diinn_conv_wino4_ws_status")
            worst = max(worst, st.value)
        return worst

    def _forward_hip_trunk(self, shallow):
        import ctypes as C
        from . import _native
        lib = _native.load()
        b, _, h, w = shallow.shape
        shallow = shallow.contiguous()
        dev = shallow.device
        packed, biases = self._hip_packed(dev)
        x3 = self.hip_winograd and self.hip_split_bf16
        w4 = self.hip_winograd and self.hip_winograd4 and not x3 and bool(lib.diinn_rdn_wino4_applies(b, h, w))
        # the F(2x2) image only where a kernel reads it (~0.6 GB of weight copies for an 88 MB encoder otherwise)
        needs_wino = self.hip_winograd and not w4 and (x3 or b * h * w >= _native.debug_get("DIINN_ENC_WINO_MIN"))
        algo = (_native.RDN_ALGO_X3 if x3 else _native.RDN_ALGO_WINO4 if w4 else _native.RDN_ALGO_WINO if needs_wino
                else _native.RDN_ALGO_DIRECT)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        packed_wino = self._hip_packed_wino(dev) if needs_wino else None
        packed_w4 = self._hip_packed_wino4(dev) if w4 else None
        packed_x3 = self._hip_packed_x3(dev) if x3 else None
        planes = torch.empty(lib.diinn_rdn_planes_floats(algo, b, h, w), dtype=torch.float32, device=dev)
        area = self._w4_area(dev) if w4 else None
        out = torch.empty_like(shallow)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _native.check(lib.diinn_rdn_forward_ex(stream, algo, ptr(shallow), ptr(packed), ptr(packed_wino), ptr(packed_w4),
                                                   ptr(packed_x3), ptr(biases), ptr(planes), ptr(area), ptr(out), b, h, w),
                          "diinn_rdn_forward_ex")
        return out

    def _sfe1_hip(self, x):
        """SFENet1 on the library's own kernel (diinn_sfe1_forward): the whole inference encoder then runs without a
        library convolution."""
        import ctypes as C
        from . import _native
        lib = _native.load()
        x = x.contiguous()
        b, c, h, w = x.shape
        wt, bias = self.SFENet1.weight.detach().contiguous(), self.SFENet1.bias.detach().contiguous()
        out = torch.empty((b, 64, h, w), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _native.check(lib.diinn_sfe1_forward(stream, C.c_void_p(x.data_ptr()), c, C.c_void_p(wt.data_ptr()),
                                                 C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), b, h, w),
                          "diinn_sfe1_forward")
        return out

    def forward(self, x):
        if (self._hip_ok and self.hip_trunk_max_pixels is not None and x.is_cuda and not torch.is_grad_enabled()
                and x.dtype == torch.float32 and x.shape[0] * x.shape[-2] * x.shape[-1] <= self.hip_trunk_max_pixels):
            if x.shape[1] <= 4 and self.SFENet1.weight.dtype == torch.float32:
                return self._forward_hip_trunk(self._sfe1_hip(x))
            return self._forward_hip_trunk(self.SFENet1(x))
        shallow = self.SFENet1(x)
        x = self.SFENet2(shallow)
        blocks = []
        for rdb in self.RDBs:
            x = rdb(x)
            blocks.append(x)
        return self.GFF(torch.cat(blocks, 1)) + shallow


def make_rdn(G0=64, RDNkSize=3, RDNconfig="B", scale=2, no_upsampling=True):
    if not no_upsampling:
        raise NotImplementedError("DIINN uses RDN as a feature encoder only (no_upsampling=True)")
    return RDN(G0=G0, RDNkSize=RDNkSize, RDNconfig=RDNconfig)


# ---------------------------------------------------------------------------
# model assemblies
# ---------------------------------------------------------------------------
class _GraphReplay:
    """hipGraph replay of a whole inference forward (encoder convolutions + decoder kernels), one graph per
    (input shape, output size, parameter versions).  Small inputs are launch-bound in the encoder (48x48:
    ~7 ms eager for ~150 launches); the graph removes that.  The decoder's C-ABI launches never allocate
    or synchronise, so they capture as they are."""

    MAX_GRAPHS = 8

    def _init_graphs(self, graphs: bool):
        self.graphs = graphs
        self._graph_cache: Dict[Any, Any] = {}

    def _use_graph(self, x) -> bool:
        return self.graphs and x.is_cuda and not torch.is_grad_enabled()

    def _graph_key(self, x, size):
        """Everything that decides which kernels a capture records and which buffers they read: input geometry,
        output size, the identity and version of every parameter (a re-assigned ``param.data`` changes the
        pointer, an in-place update the version), and the knobs that switch code paths."""
        dec = getattr(self, "decoder", None)
        enc = getattr(self, "encoder", None)
        return (tuple(x.shape), x.dtype, x.device, int(size[0]), int(size[1]),
                tuple((p.data_ptr(), p._version) for p in self.parameters()),
                getattr(dec, "sin_mode", None), getattr(dec, "compute", None), getattr(dec, "mode", None),
                getattr(enc, "hip_trunk_max_pixels", None), getattr(enc, "hip_winograd", None),
                getattr(enc, "hip_split_bf16", None), getattr(enc, "hip_winograd4", None))

    def _forward_graphed(self, x, size, bsize):
        key = self._graph_key(x, size)
        entry = self._graph_cache.get(key)
        if entry is None:
            if len(self._graph_cache) >= self.MAX_GRAPHS:
                self._graph_cache.pop(next(iter(self._graph_cache)))
            static_x = x.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up outside capture (MIOpen find, weight packing)
                for _ in range(2):
                    self._forward_eager(static_x, size, bsize)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                # every buffer the captured kernels touch is either allocated inside the capture (the graph's
                # private pool: workspaces, activations, the result) or kept alive by the entry below (packed
                # weight images, which live in module caches that a later call may replace)
                static_y = self._forward_eager(static_x, size, bsize)
            keep = [getattr(m, a, None) for m in self.modules() for a in ("_packed", "_hip_pack", "_hip_x3", "_hip_wu4", "_hip_wu2")]
            entry = (graph, static_x, static_y, keep)
            self._graph_cache[key] = entry
        graph, static_x, static_y = entry[:3]
        static_x.copy_(x)
        graph.replay()
        return static_y.clone()


class DIINN(nn.Module, _GraphReplay):
    """Reference diinn.py:8-19: ``decoder(encoder(x), size, bsize)``.  ``graphs=True`` (inference only)
    replays the whole forward from a hipGraph (see ``_GraphReplay``)."""

    def __init__(self, mode, init_q, graphs: bool = False):
        super().__init__()
        self.encoder = make_rdn()
        self.decoder = ImplicitDecoder(mode=mode, init_q=init_q)
        self._init_graphs(graphs)

    def _forward_eager(self, x, size, bsize=None):
        return self.decoder(self.encoder(x), size, bsize)

    def set_split_bf16(self, enabled: bool = True) -> "DIINN":
        """Switch both optional split-bf16 modes (not in the reference; DESIGN.md sections 3.5 and 3.9): the decoder's
        per-pixel layers (``decoder.compute = "bf16x3"``) and the encoder's 3x3 layers (``encoder.hip_split_bf16``).  The
        output stays within the reference tolerance (1e-4 x max(1, |ref|)); 256x256 x4: 17.9 -> 10.1 ms per forward."""
        self.decoder.compute = "bf16x3" if enabled else "f32"
        self.encoder.hip_split_bf16 = bool(enabled)
        return self

    def forward(self, x, size, bsize=None):
        if self._use_graph(x):
            return self._forward_graphed(x, size, bsize)
        return self._forward_eager(x, size, bsize)

    @torch.no_grad()
    def forward_sharded(self, x, size, src: int = 0, gather_to: Optional[int] = 0, group=None, mode: str = "halo"):
        """The same forward with the HR grid cut into row bands, one per rank of the process group (one process per
        GPU under ``torch.distributed``): rank ``src`` runs the encoder, every rank receives the
        LR feature rows its band reads (+ a one-row halo) and decodes its band, and the image is assembled on
        ``gather_to`` (returned there, ``None`` elsewhere) -- or, with ``gather_to=None``, every rank gets
        ``(band, (y0, y1))``.  Every rank passes ``x`` (only its shape is used away from ``src``).  Inference, mode 3.
        With one rank this is ``forward``.  The reference has no multi-GPU inference (benchmarks.py:13: devices=1)."""
        import torch.distributed as dist
        from . import sharded as S
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            out = self._forward_eager(x, size, None)
            return out if gather_to is not None else (out, (0, int(size[0])))
        if self.decoder.mode != 3 or self.decoder.init_q:
            raise NotImplementedError("forward_sharded covers the mode-3 decoder")
        hu, wu = size
        b, _, h, w = x.shape
        shape = (int(b), 64, int(h), int(w))
        packed = self.decoder.packed_weights(x.device)
        # the arithmetic and the sine mode are baked into a BandDecoder at construction: they are part of the key, so
        # switching ``decoder.compute`` / ``sin_mode`` (or ``set_split_bf16``) between two calls rebuilds it
        key = (shape, int(hu), int(wu), packed.data_ptr(), src, mode, id(group),
               self.decoder.compute, self.decoder.sin_mode)
        if getattr(self, "_band_key", None) != key:           # band-sized buffers are allocated once per geometry
            self._band_dec = S.BandDecoder(shape, (int(hu), int(wu)), packed, group=group, src=src, mode=mode,
                                           sin_mode=self.decoder.sin_mode, compute=self.decoder.compute)
            self._band_key = key
        dec = self._band_dec
        feat = self.encoder(x).contiguous() if dist.get_rank(group) == src else None
        band = dec.step(feat)
        if gather_to is None:
            return band, (dec.band.y0, dec.band.y1)
        return dec.gather(band, dst=gather_to)


class MLP(nn.Module):
    """Linear/ReLU stack with the reference's parameter names ``layers.{0,2,4,...}`` (mlp.py:3-20)."""

    def __init__(self, in_dim, out_dim, hidden_list):
        super().__init__()
        layers, last = [], in_dim
        for width in hidden_list:
            layers += [nn.Linear(last, width), nn.ReLU()]
            last = width
        layers.append(nn.Linear(last, out_dim))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x.reshape(-1, x.shape[-1])).view(*x.shape[:-1], -1)


class LIIF(nn.Module, _GraphReplay):
    """The LIIF comparison model (reference liif.py:9-155; Chen et al. 2021) behind the same
    ``forward(inp, size, bsize=None)`` boundary: RDN encoder (PyTorch-ROCm) + the implicit MLP decoder on
    the HIP path (``liif_kernel``).  Constructor defaults only (local ensemble, feature unfolding, cell
    decoding -- what ``make_net('liif')`` builds, sr_module.py:45-46); inference only.  ``bsize`` is the
    reference's query-chunk size, a memory knob: accepted and ignored."""

    def __init__(self, local_ensemble=True, feat_unfold=True, cell_decode=True, graphs: bool = False):
        super().__init__()
        self._init_graphs(graphs)
        if not (local_ensemble and feat_unfold and cell_decode):
            raise NotImplementedError("the HIP path implements LIIF with local_ensemble, feat_unfold and cell_decode on")
        self.local_ensemble, self.feat_unfold, self.cell_decode = local_ensemble, feat_unfold, cell_decode
        self.encoder = make_rdn()
        self.imnet = MLP(self.encoder.out_dim * 9 + 4, 3, [256, 256, 256, 256])
        self._packed = None
        self._packed_key = None

    def gen_feat(self, inp):
        return self.encoder(inp)

    def _packed_weights(self, device):
        from .decoder import pack_liif_state_dict
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.imnet.parameters())
        if self._packed is None or self._packed_key != key:
            self._packed = pack_liif_state_dict(self.imnet.state_dict(), prefix="").to(device)
            self._packed_key = key
        return self._packed

    def _forward_eager(self, inp, size, bsize=None):
        from .decoder import liif_decode_features
        feat = self.gen_feat(inp)
        return liif_decode_features(feat, self._packed_weights(feat.device), size)

    def forward(self, inp, size, bsize=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("diinn_amd: LIIF runs on the HIP path for inference only; call under torch.no_grad()")
        if self._use_graph(inp):
            return self._forward_graphed(inp, size, bsize)
        return self._forward_eager(inp, size, bsize)


class MetaSR(nn.Module, _GraphReplay):
    """The MetaSR comparison model (reference metasr.py:22-135; Hu et al. 2019 as re-implemented by LIIF)
    behind ``forward(inp, size, bsize=None)``: RDN encoder (PyTorch-ROCm) + the meta-upscale decoder on the
    HIP path (``metasr_kernel``).  Inference only; ``bsize`` accepted and ignored."""

    def __init__(self, graphs: bool = False):
        super().__init__()
        self._init_graphs(graphs)
        self.encoder = make_rdn()
        self.imnet = MLP(3, self.encoder.out_dim * 9 * 3, [256])
        self._packed = None
        self._packed_key = None

    def gen_feat(self, inp):
        self.feat = self.encoder(inp)
        return self.feat

    def _packed_weights(self, device):
        from .decoder import pack_metasr_state_dict
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.imnet.parameters())
        if self._packed is None or self._packed_key != key:
            self._packed = pack_metasr_state_dict(self.imnet.state_dict(), prefix="").to(device)
            self._packed_key = key
        return self._packed

    def _forward_eager(self, inp, size, bsize=None):
        from .decoder import metasr_decode_features
        feat = self.gen_feat(inp)
        return metasr_decode_features(feat, self._packed_weights(feat.device), size)

    def forward(self, inp, size, bsize=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("diinn_amd: MetaSR runs on the HIP path for inference only; call under torch.no_grad()")
        if self._use_graph(inp):
            return self._forward_graphed(inp, size, bsize)
        return self._forward_eager(inp, size, bsize)


class BICUBIC_NET(nn.Module):
    """Antialiased bicubic resize baseline (reference sr_module.py:53-60 via torchvision.Resize,
    which lowers to this interpolate call)."""

    def forward(self, x, size, eval_bsize=None):
        return F.interpolate(x, size=tuple(int(s) for s in size), mode="bicubic", align_corners=False, antialias=True)


def make_net(arch, mode, init_q):
    if arch == "diinn":
        return DIINN(mode=mode, init_q=init_q)
    if arch == "bicubic":
        return BICUBIC_NET()
    if arch == "liif":
        return LIIF()
    if arch == "metasr":
        return MetaSR()
    return None   # the reference's make_net falls through to None for unknown names


class SRLitModule(nn.Module):
    """Inference surface of the reference LightningModule (sr_module.py:62-125)."""

    def __init__(self, arch: str, mode: int = 1, init_q: bool = False, lr: float = 1e-4, lr_gamma: float = 0.5,
                 lr_step: int = 10, eval_bsize: int = 30000):
        super().__init__()
        self.hparams = SimpleNamespace(arch=arch, mode=mode, init_q=init_q, lr=lr, lr_gamma=lr_gamma,
                                       lr_step=lr_step, eval_bsize=eval_bsize)
        self.net = make_net(arch, mode, init_q)
        self.register_buffer("sub", torch.FloatTensor([0.5]).view(1, -1, 1, 1))
        self.register_buffer("div", torch.FloatTensor([0.5]).view(1, -1, 1, 1))
        self.criterion = nn.L1Loss()

    def forward(self, x: torch.Tensor, size, eval_bsize=None):
        return self.net(x, size, eval_bsize)

    def forward_sharded(self, x: torch.Tensor, size, **kw):
        """``forward`` with the HR grid sharded by row bands over the ranks of ``torch.distributed`` (DIINN only)."""
        if not hasattr(self.net, "forward_sharded"):
            raise NotImplementedError(f"arch {self.hparams.arch}: only DIINN has a sharded forward")
        return self.net.forward_sharded(x, size, **kw)

    def step(self, batch: Any, eval_bsize=None):
        """Boundary restatement, not new design: this loop IS reference sr_module.py:113-125 (``batch`` maps scale ->
        (lr, hr, name); normalise by sub/div, decode to the HR size, L1, de-normalise and clamp), kept statement for
        statement because ``training_step`` / ``validation_step`` / ``test_step`` and any user subclass depend on its exact
        return value ``(mean loss, {scale: prediction in [0, 1]})``."""
        loss = 0
        pred_hrs: Dict[Any, torch.Tensor] = {}
        for scale in batch:
            lr, hr, _ = batch[scale]
            lr = (lr - self.sub) / self.div
            hr = (hr - self.sub) / self.div
            pred_hr = self.forward(lr, hr.shape[-2:], eval_bsize)
            loss += self.criterion(pred_hr, hr)
            pred_hrs[scale] = (pred_hr * self.div + self.sub).clamp_(0, 1)
        return loss / len(batch), pred_hrs

    def training_step(self, batch: Any, batch_idx: int = 0):
        """sr_module.py:127-137: the decoder runs under autograd with bsize=None (training.py: HIP forward
        with saved planes + HIP backward)."""
        loss, _ = self.step(batch)
        return {"loss": loss}

    @torch.no_grad()
    def validation_step(self, batch: Any, batch_idx: int = 0):
        """sr_module.py:143-154: loss and the DIV2K-style PSNR (border shaved by the scale) per scale."""
        loss, pred_hrs = self.step(batch, self.hparams.eval_bsize)
        res = {"val/loss": loss}
        for scale in batch:
            res[f"val/psnr_x{scale}"] = calc_psnr(pred_hrs[scale], batch[scale][1], dataset="div2k", scale=scale, rgb_range=1)
        return res

    def configure_optimizers(self):
        """Boundary restatement of sr_module.py:185-194 (the return shape is Lightning's contract): Adam(lr) +
        StepLR(lr_step, lr_gamma), one scheduler step per epoch."""
        optimizer = torch.optim.Adam(self.parameters(), lr=self.hparams.lr)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer=optimizer, step_size=self.hparams.lr_step,
                                                    gamma=self.hparams.lr_gamma)
        return [optimizer], [scheduler]

    def checkpoint(self) -> Dict[str, Any]:
        """The two entries of a Lightning checkpoint that ``load_from_checkpoint`` (here and in the reference) reads."""
        return {"state_dict": self.state_dict(), "hyper_parameters": dict(vars(self.hparams))}

    @torch.no_grad()
    def test_step(self, batch: Any, batch_idx: int = 0, dataloader_idx: Optional[int] = None):
        """sr_module.py:159-180: decode with the checkpoint's eval_bsize, then PSNR, SSIM and the PSNR of
        the re-downsampled prediction against the re-downsampled target, per scale (data_range=1)."""
        _, pred_hrs = self.step(batch, self.hparams.eval_bsize)
        res = {}
        for scale in batch:
            hr = batch[scale][1]
            lr_size = (round(hr.shape[-2] / scale), round(hr.shape[-1] / scale))
            res[scale] = {
                "psnr_res": psnr(pred_hrs[scale], hr, data_range=1),
                "ssim_res": ssim(pred_hrs[scale], hr, data_range=1),
                "lr_psnr_res": psnr(resize_fn(pred_hrs[scale], lr_size), resize_fn(hr, lr_size), data_range=1),
            }
        return res

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, strict: bool = True, **overrides):
        """Lightning checkpoint dict: ``hyper_parameters`` (ctor kwargs saved by save_hyperparameters,
        sr_module.py:91) and ``state_dict`` (keys ``net.encoder.*``, ``net.decoder.*``, ``sub``, ``div``)."""
        # plain tensors + a dict of primitives load under weights_only=True; a checkpoint that pickles other
        # classes (Lightning's AttributeDict, callbacks, LightningCLI namespaces) executes code on load and needs an
        # explicit opt-in: DIINN_TRUST_CKPT=1 (README "Checkpoints")
        import os
        try:
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
        except Exception as e:
            if os.environ.get("DIINN_TRUST_CKPT") == "1":
                ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            else:
                raise RuntimeError(
                    f"{checkpoint_path} did not load as plain tensors (torch.load(weights_only=True): "
                    f"{type(e).__name__}: {e}).  Checkpoints written by Lightning / LightningCLI can pickle their own "
                    f"classes; unpickling those runs code from the file.  If you trust this file, set "
                    f"DIINN_TRUST_CKPT=1 to load it with weights_only=False.") from e
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        known = ("arch", "mode", "init_q", "lr", "lr_gamma", "lr_step", "eval_bsize")
        model = cls(**{k: v for k, v in hp.items() if k in known})
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        return model.eval()
