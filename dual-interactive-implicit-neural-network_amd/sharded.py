"""Row-band sharding of the HR grid across the GPUs of one node (one process per GPU).

Every HR pixel of the mode-3 decoder depends only on the LR features in the 3x3
neighbourhood of its nearest LR cell (all decoder convs are 1x1; the reference's
own column chunking, diinn.py:149-160, relies on the same independence).  The HR
grid is therefore cut into contiguous ROW BANDS, one per rank, with no
cross-band reduction.  The only exchange step is handing each rank the LR
feature rows its band reads (band rows + a one-row halo for the 3x3 unfold),
sent point-to-point from the rank that ran the encoder over RCCL/xGMI
(``torch.distributed`` backend "nccl"), or a plain broadcast of the whole map.

Everything a rank holds is band-sized (``BandDecoder``): a feature window
``[B,64,a1-a0,W]`` that the hand-off receives straight into, a P workspace of
the band's LR rows, the output band ``[B,3,y1-y0,Wu]``.  The staging buffers of
the sender and all receive buffers are allocated once; a step allocates nothing.
The kernels read the windows through the row-window entry points of the C ABI
(``diinn_decode_win``), so a band is bit-identical to the same rows of an
unsharded decode.

Outputs stay sharded (rank r owns HR rows ``[y0,y1)``) unless ``gather`` is
called, which assembles the image on one rank for ``demo2``-style callers.
The reference has no multi-GPU inference to mirror (benchmarks.py:13: devices=1).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

IN_CHANNELS = 64


def band_for_rank(hu: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous HR row band [y0,y1) of ``rank``; bands differ by at most one row."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(hu), world)
    y0 = rank * base + min(rank, extra)
    y1 = y0 + base + (1 if rank < extra else 0)
    return y0, y1


def all_bands(hu: int, world: int) -> List[Tuple[int, int]]:
    return [band_for_rank(hu, r, world) for r in range(world)]


def feature_rows_for_band(h: int, lr_rows: Tuple[int, int]) -> Tuple[int, int]:
    """LR feature rows [a0,a1) a band must hold: its P rows plus the 3x3 halo, clipped to the map
    (rows outside the map are the unfold's zero padding, diinn.py:168)."""
    r0, r1 = lr_rows
    return max(r0 - 1, 0), min(r1 + 1, h)


@dataclass(frozen=True)
class Band:
    """One rank's share: HR rows [y0,y1), the LR rows [r0,r1) of P it needs, the feature rows [a0,a1)
    (P rows + halo) it must hold.  An empty band (more ranks than HR rows) has y1 == y0 and holds nothing."""
    y0: int
    y1: int
    r0: int
    r1: int
    a0: int
    a1: int

    @property
    def empty(self) -> bool:
        return self.y1 <= self.y0


def plan_bands(h: int, hu: int, wu: int, world: int) -> List[Band]:
    """The band of every rank for an LR map of height ``h`` decoded to (hu, wu).  Uses the library's own
    index code (``diinn_window_rows``) so the rows match what the kernels read."""
    from . import decoder as D
    bands = []
    for (y0, y1) in all_bands(hu, world):
        if y1 <= y0:
            bands.append(Band(y0, y0, 0, 0, 0, 0))
            continue
        (a0, an), (r0, rn) = D.window_rows(h, hu, wu, y0, y1)
        bands.append(Band(y0, y1, r0, r0 + rn, a0, a0 + an))
    return bands


def _p2p(ops):
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class BandExchange:
    """The feature hand-off and the optional output gather of the row-band split, with every staging
    and receive buffer allocated once.  Transport only (no kernels): runs on any backend / device,
    which is how the gloo tests cover it on CPU."""

    def __init__(self, shape: Sequence[int], size: Sequence[int], bands: List[Band], device, group=None,
                 src: int = 0, mode: str = "halo"):
        if mode not in ("halo", "bcast"):
            raise ValueError("mode must be 'halo' or 'bcast'")
        self.shape = tuple(int(v) for v in shape)
        self.hu, self.wu = int(size[0]), int(size[1])
        self.bands = bands
        self.group = group
        self.src = src
        self.mode = mode
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if len(bands) != self.world:
            raise ValueError("one band per rank")
        self.band = bands[self.rank]
        b, c, h, w = self.shape
        dev = torch.device(device)
        self.device = dev
        self.send_stage: List[Optional[torch.Tensor]] = [None] * self.world
        self.feat_win: Optional[torch.Tensor] = None       # non-src ranks: what the hand-off fills
        if self.world > 1 and mode == "halo":
            if self.rank == src:
                for r, bd in enumerate(bands):
                    if r != src and not bd.empty:
                        self.send_stage[r] = torch.empty((b, c, bd.a1 - bd.a0, w), dtype=torch.float32, device=dev)
            elif not self.band.empty:
                self.feat_win = torch.empty((b, c, self.band.a1 - self.band.a0, w), dtype=torch.float32, device=dev)
        elif self.world > 1 and self.rank != src:           # bcast: the whole map everywhere
            self.feat_win = torch.empty(self.shape, dtype=torch.float32, device=dev)

    # -- features: src -> every rank ----------------------------------------------------------
    def handoff(self, feat: Optional[torch.Tensor]) -> Tuple[Optional[torch.Tensor], int]:
        """Returns (feature window, first LR row it holds).  On ``src`` that is the full map itself (row 0);
        elsewhere the pre-allocated window, filled by this call.  Allocates nothing."""
        if self.rank == self.src:
            if feat is None or tuple(feat.shape) != self.shape:
                raise ValueError("src rank must pass the full feature map")
        if self.world == 1:
            return feat, 0
        if self.mode == "bcast":
            buf = feat if self.rank == self.src else self.feat_win
            dist.broadcast(buf, src=self.src, group=self.group)
            return buf, 0
        ops = []
        if self.rank == self.src:
            for r, bd in enumerate(self.bands):
                stage = self.send_stage[r]
                if stage is None:
                    continue
                stage.copy_(feat[:, :, bd.a0:bd.a1, :])    # strided rows -> the contiguous message
                ops.append(dist.P2POp(dist.isend, stage, r, self.group))
            _p2p(ops)
            return feat, 0
        if self.band.empty:
            return None, 0
        ops.append(dist.P2POp(dist.irecv, self.feat_win, self.src, self.group))
        _p2p(ops)
        return self.feat_win, self.band.a0

    # -- output bands -> one rank ---------------------------------------------------------------
    def gather(self, out_band: Optional[torch.Tensor], dst: int = 0,
               out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Assemble the [B,3,Hu,Wu] image on ``dst`` from every rank's band [B,3,y1-y0,Wu].  Each (batch,
        channel) plane of a band is contiguous on both sides, so the bands are received in place (no
        staging copy).  Returns the image on ``dst`` and None elsewhere."""
        b = self.shape[0]
        if self.rank == dst:
            if out is None:
                out = torch.empty((b, 3, self.hu, self.wu), dtype=torch.float32, device=self.device)
            if not self.band.empty:
                out[:, :, self.band.y0:self.band.y1, :].copy_(out_band)
        if self.world == 1:
            return out
        ops = []
        if self.rank == dst:
            for r, bd in enumerate(self.bands):
                if r == dst or bd.empty:
                    continue
                for bi in range(b):
                    for ch in range(3):
                        ops.append(dist.P2POp(dist.irecv, out[bi, ch, bd.y0:bd.y1, :], r, self.group))
        elif not self.band.empty:
            for bi in range(b):
                for ch in range(3):
                    ops.append(dist.P2POp(dist.isend, out_band[bi, ch], dst, self.group))
        _p2p(ops)
        return out if self.rank == dst else None


class BandDecoder(BandExchange):
    """One rank of the sharded decode: the exchange above plus the band-sized P workspace and output band,
    allocated once, and the HIP kernels run through the row-window C ABI."""

    def __init__(self, shape: Sequence[int], size: Sequence[int], packed: torch.Tensor, group=None, src: int = 0,
                 mode: str = "halo", sin_mode: Optional[int] = None, compute: str = "f32"):
        from . import _native
        b, c, h, w = (int(v) for v in shape)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        bands = plan_bands(h, int(size[0]), int(size[1]), world)
        super().__init__(shape, size, bands, packed.device, group=group, src=src, mode=mode)
        self.packed = packed
        self.sin_mode = _native.SIN_DEFAULT if sin_mode is None else sin_mode
        self.compute = compute
        bd = self.band
        self.p_win = None if bd.empty else torch.empty(b * (bd.r1 - bd.r0) * w * 1024, dtype=torch.float32,
                                                       device=packed.device)
        self.out_band = None if bd.empty else torch.empty((b, 3, bd.y1 - bd.y0, self.wu), dtype=torch.float32,
                                                          device=packed.device)

    def decode_local(self, feat_win: torch.Tensor, feat_row0: int) -> Optional[torch.Tensor]:
        """P for the band's LR rows + the decode kernel over the band, from a feature window."""
        from . import decoder as D
        bd = self.band
        if bd.empty:
            return None
        return D.decode_window(feat_win, feat_row0, self.shape[2], self.packed, (self.hu, self.wu), (bd.y0, bd.y1),
                               p_win=self.p_win, out_win=self.out_band, sin_mode=self.sin_mode, compute=self.compute)

    def step(self, feat: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """One sharded decode: hand-off from ``src``, then this rank's band.  Returns the band
        [B,3,y1-y0,Wu] (None for an empty band)."""
        win, row0 = self.handoff(feat)
        return self.decode_local(win, row0)


def decode_sharded(feat: Optional[torch.Tensor], shape: Tuple[int, int, int, int], packed: torch.Tensor,
                   size, src: int = 0, group=None, mode: str = "halo", sin_mode: Optional[int] = None,
                   compute: str = "f32", gather_to: Optional[int] = None):
    """One-shot convenience wrapper (allocates its buffers; loops should keep a ``BandDecoder``).
    Returns (band, (y0, y1)) -- or (image on ``gather_to`` / None elsewhere, (y0, y1)) with ``gather_to``."""
    dec = BandDecoder(shape, size, packed, group=group, src=src, mode=mode, sin_mode=sin_mode, compute=compute)
    band = dec.step(feat)
    rows = (dec.band.y0, dec.band.y1)
    if gather_to is not None:
        return dec.gather(band, dst=gather_to), rows
    return band, rows
