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
(``diinn_decode_win``).  In fp32 a band is bit-identical to the same rows of an
unsharded decode (the P form never depends on the band and the latency /
throughput decode kernels are bit-equal); the optional bf16 modes pick their
kernel from the full image's geometry, not the band's, for the same reason
(``launch_decode_bf16``), so their bands stitch exactly too.

The source rank's sends run on a side stream while it decodes its own band
(``handoff`` starts them, ``complete`` orders the caller's stream behind them):
rank 0 is the critical path of a step, it must not wait for its peers' data.

Outputs stay sharded (rank r owns HR rows ``[y0,y1)``) unless ``gather`` is
called, which assembles the image on one rank for ``demo2``-style callers.
The reference has no multi-GPU inference to mirror (benchmarks.py:13: devices=1).
"""
from __future__ import annotations

import contextlib
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


def _post(ops):
    """Post point-to-point operations (one coalesced group on RCCL); returns the requests."""
    return list(dist.batch_isend_irecv(ops)) if ops else []


def _wait(reqs):
    for req in reqs:
        req.wait()


class BandExchange:
    """The feature hand-off and the optional output gather of the row-band split, with every staging
    and receive buffer allocated once.  Transport only (no kernels).

    Three transports, one control flow:
    * ROCm device + RCCL (backend "nccl"): device buffers go on the wire as they are.  The source rank issues its
      stage copies and sends on a SIDE stream and returns at once, so its own band (the critical path: ``value``
      is the max over ranks) runs concurrently with the sends; ``complete()`` orders the caller's stream
      behind them.
    * CPU tensors + gloo: the same flow without streams (``tests/test_sharding_gloo.py``).
    * ROCm device + gloo (``host_staged``): gloo has no device point-to-point, so messages pass through pinned
      host buffers.  This is the one-GPU test transport of ``bench.py --backend gloo`` (RCCL refuses two ranks
      on one device); the control flow -- side stream, deferred completion, band-sized windows -- is the same.
    """

    def __init__(self, shape: Sequence[int], size: Sequence[int], bands: List[Band], device, group=None,
                 src: int = 0, mode: str = "halo", solo: bool = False):
        if mode not in ("halo", "bcast"):
            raise ValueError("mode must be 'halo' or 'bcast'")
        self.shape = tuple(int(v) for v in shape)
        self.hu, self.wu = int(size[0]), int(size[1])
        self.bands = bands
        self.group = group
        self.src = src
        self.mode = mode
        live = dist.is_initialized() and not solo           # solo: one rank decodes alone inside a larger job
        self.rank = dist.get_rank(group) if live else 0
        self.world = dist.get_world_size(group) if live else 1
        if len(bands) != self.world:
            raise ValueError("one band per rank")
        self.band = bands[self.rank]
        b, c, h, w = self.shape
        dev = torch.device(device)
        self.device = dev
        backend = dist.get_backend(group) if live else None
        self.host_staged = dev.type == "cuda" and self.world > 1 and backend == "gloo"
        self.side = torch.cuda.Stream(device=dev) if dev.type == "cuda" and self.world > 1 else None
        self._pending = []                                   # src: sends of the running step
        self._staged = None                                  # src, host_staged: event behind the host copies
        self._recv_done = None                               # host_staged receivers: event behind the last host -> device copy
        self._gather_done = None                             # host_staged gather root: the same for the gather stages
        self.send_stage: List[Optional[torch.Tensor]] = [None] * self.world
        self.send_host: List[Optional[torch.Tensor]] = [None] * self.world
        self.feat_win: Optional[torch.Tensor] = None        # non-src ranks: what the hand-off fills
        self.recv_host: Optional[torch.Tensor] = None
        self.gather_stage: List[Optional[torch.Tensor]] = [None] * self.world
        self.gather_host: Optional[torch.Tensor] = None

        def host(shape_):
            return torch.empty(shape_, dtype=torch.float32, pin_memory=True)

        if self.world > 1 and mode == "halo":
            if self.rank == src:
                for r, bd in enumerate(bands):
                    if r != src and not bd.empty:
                        self.send_stage[r] = torch.empty((b, c, bd.a1 - bd.a0, w), dtype=torch.float32, device=dev)
                        if self.host_staged:
                            self.send_host[r] = host((b, c, bd.a1 - bd.a0, w))
            elif not self.band.empty:
                self.feat_win = torch.empty((b, c, self.band.a1 - self.band.a0, w), dtype=torch.float32, device=dev)
                if self.host_staged:
                    self.recv_host = host(self.feat_win.shape)
        elif self.world > 1:                                 # bcast: the whole map everywhere
            if self.rank != src:
                self.feat_win = torch.empty(self.shape, dtype=torch.float32, device=dev)
            if self.host_staged:
                self.recv_host = host(self.shape)

    # -- features: src -> every rank ----------------------------------------------------------
    def handoff(self, feat: Optional[torch.Tensor]) -> Tuple[Optional[torch.Tensor], int]:
        """Returns (feature window, first LR row it holds).  On ``src`` that is the full map itself (row 0) and the
        sends are only STARTED: call ``complete()`` after queueing this rank's own work.  Elsewhere the
        pre-allocated window, filled by this call (stream-ordered on a device).  Allocates nothing."""
        if self.rank == self.src:
            if feat is None or tuple(feat.shape) != self.shape:
                raise ValueError("src rank must pass the full feature map")
        if self.world == 1:
            return feat, 0
        if self.mode == "bcast":
            return self._broadcast(feat), 0
        if self.rank == self.src:
            self.complete()                                  # the stages are free again (no-op in a step loop)
            cur = torch.cuda.current_stream(self.device) if self.side is not None else None
            if self.side is not None:
                self.side.wait_stream(cur)                   # the features must be final before they are copied
            with (torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()):
                ops = []
                for r, bd in enumerate(self.bands):
                    stage = self.send_stage[r]
                    if stage is None:
                        continue
                    stage.copy_(feat[:, :, bd.a0:bd.a1, :])    # strided rows -> the contiguous message
                    if self.host_staged:
                        self.send_host[r].copy_(stage, non_blocking=True)
                    else:
                        ops.append(dist.P2POp(dist.isend, stage, r, self.group))
                if self.host_staged:
                    self._staged = torch.cuda.Event()
                    self._staged.record(self.side)
                else:
                    self._pending = _post(ops)
                    if self.side is not None:
                        _wait(self._pending)                 # RCCL: orders the side stream behind the sends
            return feat, 0
        if self.band.empty:
            return None, 0
        if self.host_staged:
            # the previous step's asynchronous host -> device copy may still be reading the pinned message: it must
            # have finished before the next message is received into the same buffer
            if self._recv_done is not None:
                self._recv_done.synchronize()
            _wait(_post([dist.P2POp(dist.irecv, self.recv_host, self.src, self.group)]))
            self.feat_win.copy_(self.recv_host, non_blocking=True)
            self._recv_done = torch.cuda.Event()
            self._recv_done.record()
        else:
            _wait(_post([dist.P2POp(dist.irecv, self.feat_win, self.src, self.group)]))
        return self.feat_win, self.band.a0

    def complete(self) -> None:
        """Finish the hand-off this rank started: afterwards work queued on the current stream (or, on CPU, the
        caller itself) is ordered behind the sends, so the feature map may be overwritten.  A rank that only
        receives has nothing to finish."""
        if self._staged is not None:                         # host-staged: the sends start here, after the caller
            self._staged.synchronize()                       # has queued its own kernels
            self._staged = None
            self._pending = _post([dist.P2POp(dist.isend, self.send_host[r], r, self.group)
                                   for r in range(self.world) if self.send_host[r] is not None])
        if self._pending:
            if self.side is None or self.host_staged:
                _wait(self._pending)
            self._pending = []
        if self.side is not None and not self.host_staged:
            torch.cuda.current_stream(self.device).wait_stream(self.side)

    def _broadcast(self, feat):
        buf = feat if self.rank == self.src else self.feat_win
        if self.host_staged:
            if self._recv_done is not None:                  # the last step's copy out of the pinned message (see handoff)
                self._recv_done.synchronize()
            if self.rank == self.src:
                self.recv_host.copy_(feat)
            dist.broadcast(self.recv_host, src=self.src, group=self.group)
            if self.rank != self.src:
                buf.copy_(self.recv_host, non_blocking=True)
                self._recv_done = torch.cuda.Event()
                self._recv_done.record()
        else:
            dist.broadcast(buf, src=self.src, group=self.group)
        return buf

    # -- output bands -> one rank ---------------------------------------------------------------
    def gather(self, out_band: Optional[torch.Tensor], dst: int = 0,
               out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Assemble the [B,3,Hu,Wu] image on ``dst`` from every rank's band [B,3,y1-y0,Wu]: ONE message per
        rank (a band is contiguous on the sender), received into a band-shaped stage allocated on first use and
        placed by one strided copy.  Returns the image on ``dst`` and None elsewhere."""
        b = self.shape[0]
        self.complete()
        if self.rank == dst:
            if out is None:
                out = torch.empty((b, 3, self.hu, self.wu), dtype=torch.float32, device=self.device)
            if not self.band.empty:
                out[:, :, self.band.y0:self.band.y1, :].copy_(out_band)
        if self.world == 1:
            return out
        pin = self.host_staged
        if self.rank == dst:
            if self._gather_done is not None:                # the previous gather's copies out of the pinned stages
                self._gather_done.synchronize()
            ops, srcs = [], []
            for r, bd in enumerate(self.bands):
                if r == dst or bd.empty:
                    continue
                if self.gather_stage[r] is None:
                    self.gather_stage[r] = torch.empty((b, 3, bd.y1 - bd.y0, self.wu), dtype=torch.float32,
                                                       device="cpu" if pin else self.device, pin_memory=pin)
                ops.append(dist.P2POp(dist.irecv, self.gather_stage[r], r, self.group))
                srcs.append((r, bd))
            _wait(_post(ops))
            for r, bd in srcs:
                out[:, :, bd.y0:bd.y1, :].copy_(self.gather_stage[r], non_blocking=True)
            if pin:
                self._gather_done = torch.cuda.Event()
                self._gather_done.record()
        elif not self.band.empty:
            msg = out_band
            if pin:
                if self.gather_host is None:
                    self.gather_host = torch.empty(out_band.shape, dtype=torch.float32, pin_memory=True)
                self.gather_host.copy_(out_band)             # blocking copy: the message is complete on return
                msg = self.gather_host
            _wait(_post([dist.P2POp(dist.isend, msg.contiguous(), dst, self.group)]))
        return out if self.rank == dst else None


class BandDecoder(BandExchange):
    """One rank of the sharded decode: the exchange above plus the band-sized P workspace and output band,
    allocated once, and the HIP kernels run through the row-window C ABI.  ``solo=True`` plans a single band
    (the whole image on this rank) whatever the size of the job: ``bench.py``'s same-run one-GPU reference."""

    def __init__(self, shape: Sequence[int], size: Sequence[int], packed: torch.Tensor, group=None, src: int = 0,
                 mode: str = "halo", sin_mode: Optional[int] = None, compute: str = "f32", solo: bool = False):
        from . import _native
        b, c, h, w = (int(v) for v in shape)
        world = dist.get_world_size(group) if dist.is_initialized() and not solo else 1
        bands = plan_bands(h, int(size[0]), int(size[1]), world)
        super().__init__(shape, size, bands, packed.device, group=group, src=src, mode=mode, solo=solo)
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
        """One sharded decode: hand-off from ``src`` (started, not awaited), this rank's band queued behind it,
        then the hand-off completed.  Returns the band [B,3,y1-y0,Wu] (None for an empty band)."""
        win, row0 = self.handoff(feat)
        out = self.decode_local(win, row0)
        self.complete()
        return out


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
