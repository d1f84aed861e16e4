"""Tile sharding of the HR grid across the GPUs of one node (one process per GPU).

Every HR pixel of the mode-3 decoder depends only on the LR features in the 3x3
neighbourhood of its nearest LR cell (all decoder convs are 1x1; the reference's
own column chunking, diinn.py:149-160, relies on the same independence).  The HR
grid is therefore cut into contiguous ROW BANDS, one per rank, with no
cross-band reduction.  The only exchange step is handing each rank the LR
feature rows its band reads (band rows + a one-row halo for the 3x3 unfold),
sent point-to-point from the rank that ran the encoder over RCCL/xGMI
(``torch.distributed`` backend "nccl"), or a plain broadcast of the whole map.

Outputs stay sharded: rank r owns ``out[:, :, y0:y1, :]``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

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


def distribute_features(feat: Optional[torch.Tensor], shape: Tuple[int, int, int, int],
                        rows_per_rank: List[Tuple[int, int]], src: int = 0, group=None,
                        mode: str = "halo", device=None, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Give every rank the LR feature rows it needs, inside a full-size [B,64,H,W] buffer.

    feat          : the full feature map on rank ``src`` (None elsewhere)
    rows_per_rank : [a0,a1) LR rows needed by each rank (``feature_rows_for_band``)
    mode "halo"   : point-to-point, each rank receives only its rows (1/world of the bytes per link)
    mode "bcast"  : one broadcast of the whole map
    Returns the local full-size buffer; rows outside [a0,a1) are unspecified."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    b, c, h, w = shape
    if rank == src:
        if feat is None or tuple(feat.shape) != tuple(shape):
            raise ValueError("src rank must pass the full feature map")
        local = feat
        device = feat.device
    else:
        if buf is not None:
            local = buf
        else:
            local = torch.empty(shape, dtype=torch.float32, device=device)
    if world == 1:
        return local
    if mode == "bcast":
        dist.broadcast(local, src=src, group=group)
        return local
    if mode != "halo":
        raise ValueError("mode must be 'halo' or 'bcast'")
    ops = []
    stage = None
    if rank == src:
        keep = []
        for r in range(world):
            if r == src:
                continue
            a0, a1 = rows_per_rank[r]
            chunk = feat[:, :, a0:a1, :].contiguous()
            keep.append(chunk)
            ops.append(dist.P2POp(dist.isend, chunk, r, group))
    else:
        a0, a1 = rows_per_rank[rank]
        stage = torch.empty((b, c, a1 - a0, w), dtype=torch.float32, device=local.device)
        ops.append(dist.P2POp(dist.irecv, stage, src, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if stage is not None:
        a0, a1 = rows_per_rank[rank]
        local[:, :, a0:a1, :].copy_(stage)
    return local


def decode_sharded(feat: Optional[torch.Tensor], shape: Tuple[int, int, int, int], packed: torch.Tensor,
                   size, src: int = 0, group=None, mode: str = "halo",
                   out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                   feat_buf: Optional[torch.Tensor] = None, sin_mode: Optional[int] = None):
    """One sharded decode: distribute features from ``src``, then each rank decodes its HR band
    with the HIP kernels.  Returns (out, (y0, y1)); only out[:, :, y0:y1, :] is valid on this rank."""
    from . import decoder as D
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    b, c, h, w = shape
    hu, wu = int(size[0]), int(size[1])
    bands = all_bands(hu, world)
    need = []
    for (y0, y1) in bands:
        # an empty band (more ranks than HR rows) still takes part in the exchange with one row
        need.append(feature_rows_for_band(h, D.lr_rows_for_band(h, hu, wu, y0, y1)) if y1 > y0 else (0, 1))
    if world > 1:
        local = distribute_features(feat, shape, need, src=src, group=group, mode=mode,
                                    device=packed.device, buf=feat_buf)
    else:
        local = feat
    y0, y1 = bands[rank]
    if y1 <= y0:                     # more ranks than HR rows: nothing to decode here
        return out, (y0, y1)
    from . import _native
    out = D.decode_features(local, packed, (hu, wu), out=out, workspace=workspace, rows=(y0, y1),
                            sin_mode=_native.SIN_DEFAULT if sin_mode is None else sin_mode)
    return out, (y0, y1)
