"""CPU, world_size 2 and 4 over gloo: the multi-GPU plumbing (band partition, feature hand-off into
band-sized windows, output gather).  The per-band decode itself is the HIP kernel (GPU tests); here the
oracle stands in for it so the stitched result can be compared with the unsharded decode."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import diinn_amd.synth as synth


def test_band_partition_covers_grid():
    import diinn_amd.sharded as S
    for hu in (1, 7, 96, 1024, 8192, 2376):
        for world in (1, 2, 3, 4, 8):
            bands = S.all_bands(hu, world)
            assert bands[0][0] == 0 and bands[-1][1] == hu
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
            sizes = [b - a for a, b in bands]
            assert max(sizes) - min(sizes) <= 1


def test_feature_rows_include_halo():
    import diinn_amd.sharded as S
    assert S.feature_rows_for_band(256, (0, 64)) == (0, 65)
    assert S.feature_rows_for_band(256, (64, 128)) == (63, 129)
    assert S.feature_rows_for_band(256, (192, 256)) == (191, 256)


def test_plan_bands_matches_index_tables():
    """plan_bands (library index code) against the oracle's nearest-exact tables: every band's P rows are
    exactly the cells its HR rows select, the feature rows add the clipped 3x3 halo."""
    import diinn_oracle as orc
    import diinn_amd.sharded as S
    for (h, hu, wu, world) in [(24, 79, 66, 4), (256, 1024, 1024, 8), (720, 2376, 4224, 8), (1024, 8192, 8192, 8),
                               (5, 3, 9, 4), (40, 132, 185, 3)]:
        idx, _ = orc.axis_tables(h, hu, orc.uses_small_output_kernel(hu, wu))
        bands = S.plan_bands(h, hu, wu, world)
        assert len(bands) == world
        for bd in bands:
            if bd.empty:
                continue
            assert bd.r0 == idx[bd.y0] and bd.r1 == idx[bd.y1 - 1] + 1
            assert bd.a0 == max(bd.r0 - 1, 0) and bd.a1 == min(bd.r1 + 1, h)
        assert sum(1 for bd in bands if bd.empty) == max(0, world - hu)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, geom, q):
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import diinn_oracle as orc
    import diinn_amd.sharded as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        b, h, w, hu, wu = geom
        shape = (b, 64, h, w)
        sd = synth.decoder_state_dict(9)
        feat_np = synth.encoder_features(9, b, h, w)
        feat = torch.from_numpy(feat_np) if rank == 0 else None
        bands = S.plan_bands(h, hu, wu, world)
        ex = S.BandExchange(shape, (hu, wu), bands, "cpu", src=0, mode=mode)
        if ex.feat_win is not None:
            ex.feat_win.fill_(float("nan"))
        bd = ex.band
        ok_rows, band = True, None
        for it in range(2):                       # twice: the pre-allocated buffers are reused
            win, row0 = ex.handoff(feat)
            if rank == 0 and mode == "halo" and world > 1:
                # the source only STARTS its sends: it decodes its own band below while they are in flight
                assert len(ex._pending) == sum(1 for r, b2 in enumerate(bands) if r != 0 and not b2.empty)
            if bd.empty:
                ex.complete()
                continue
            if mode == "halo" and rank != 0:
                assert win.shape[2] == bd.a1 - bd.a0 and row0 == bd.a0       # band-sized, not the whole map
            lo = bd.a0 - row0
            ok_rows = ok_rows and bool(np.array_equal(win[:, :, lo:lo + bd.a1 - bd.a0].numpy(), feat_np[:, :, bd.a0:bd.a1]))
            # decode the band from ONLY the rows this rank holds -> must equal the same rows of the full decode
            band = orc.decode_reference_form(sd, win[:, :, lo:lo + bd.a1 - bd.a0].contiguous(), (hu, wu), None,
                                             row_range=(bd.y0, bd.y1), feat_row0=bd.a0, full_h=h)
            ex.complete()
            assert not ex._pending
        img = ex.gather(band, dst=0)
        if rank == 0 and world > 1:                # one band-shaped message per sending rank, placed by one copy
            assert sum(1 for g in ex.gather_stage if g is not None) == sum(1 for r, b2 in enumerate(bands)
                                                                            if r != 0 and not b2.empty)
        if rank == 0:
            full = orc.decode_reference_form(sd, feat_np, (hu, wu), None).numpy()
            q.put((ok_rows, float(np.abs(img.numpy() - full).max())))
        else:
            assert img is None
            q.put((ok_rows, 0.0))
    finally:
        dist.destroy_process_group()


def _run(world, mode, geom, target=_worker, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, mode, geom, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("mode", ["halo", "bcast"])
def test_two_rank_feature_handoff_and_stitch(mode):
    res = _run(2, mode, (1, 24, 20, 79, 66))
    assert all(r[0] for r in res)
    assert max(r[1] for r in res) <= 1e-6


def test_four_ranks_uneven_bands_batch2():
    """world_size 4, HR height 79 (bands of 20,20,20,19 rows), non-integer scale, batch 2."""
    res = _run(4, "halo", (2, 24, 20, 79, 66))
    assert all(r[0] for r in res)
    assert max(r[1] for r in res) <= 1e-6


def test_more_ranks_than_rows():
    """4 ranks, 3 HR rows: the last rank's band is empty and it simply sits the exchange out."""
    res = _run(4, "halo", (1, 5, 6, 3, 9))
    assert all(r[0] for r in res)
    assert max(r[1] for r in res) <= 1e-6


def test_eight_ranks_c4_and_c5_proportions():
    """world_size 8 -- the target machine -- over gloo: x8 with even bands (c4's proportions: 256 HR rows / 8) and x3.3
    with 297 = 2376 / 8 HR rows (c5's proportions: bands of 38,37,... rows whose LR windows overlap by the halo)."""
    for geom in [(1, 32, 24, 256, 192), (1, 90, 20, 297, 66)]:
        res = _run(8, "halo", geom, timeout=600)
        assert len(res) == 8 and all(r[0] for r in res)
        assert max(r[1] for r in res) <= 1e-6


def test_eight_ranks_bcast_and_more_ranks_than_rows():
    res = _run(8, "bcast", (1, 12, 10, 40, 33), timeout=600)
    assert all(r[0] for r in res) and max(r[1] for r in res) <= 1e-6
    res = _run(8, "halo", (1, 5, 6, 5, 9), timeout=600)           # 5 HR rows on 8 ranks: three empty bands
    assert all(r[0] for r in res) and max(r[1] for r in res) <= 1e-6


def test_plan_bands_at_the_baseline_sizes_for_eight_ranks():
    """plan_bands at the real c3 / target / c4 / c5 sizes for 2, 4 and 8 ranks (index math only): the bands tile the HR
    grid, pixel counts differ by < 1 %, the P rows of neighbouring bands overlap by at most one LR row (a cell whose
    HR rows straddle the cut) and together cover the map, every feature window is the P rows + the clipped halo."""
    import diinn_amd.sharded as S
    for (h, w, hu, wu) in [(512, 512, 2048, 2048), (1024, 1024, 4096, 4096), (1024, 1024, 8192, 8192),
                           (720, 1280, 2376, 4224)]:
        for world in (2, 4, 8):
            bands = S.plan_bands(h, hu, wu, world)
            assert bands[0].y0 == 0 and bands[-1].y1 == hu and bands[0].r0 == 0 and bands[-1].r1 == h
            px = [(b.y1 - b.y0) * wu for b in bands]
            assert (max(px) - min(px)) / max(px) < 0.01, (h, hu, world, px)
            for a, b in zip(bands, bands[1:]):
                assert a.y1 == b.y0 and 0 <= a.r1 - b.r0 <= 1, (a, b)
            for b in bands:
                assert (b.a0, b.a1) == (max(b.r0 - 1, 0), min(b.r1 + 1, h))
            # a rank's share of the feature map: 1/world of the rows + at most 3 (halo on both sides + a shared cell row)
            assert max(b.a1 - b.a0 for b in bands) <= -(-h // world) + 3


def _gpu_worker(rank, world, port, mode, geom, q):
    """Two ranks sharing cuda:0 (the test box has one GPU, and RCCL refuses two ranks on one device): the
    hand-off and the gather travel over gloo on CPU buffers, each rank's band is decoded by the HIP
    kernels from its band-sized window (diinn_decode_win) and compared bit-for-bit with the same rows of
    an unsharded decode."""
    import diinn_amd.decoder as D
    import diinn_amd.sharded as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        b, h, w, hu, wu = geom
        shape = (b, 64, h, w)
        packed = D.pack_state_dict(synth.decoder_state_dict(21)).to(dev)
        feat_cpu = torch.from_numpy(synth.encoder_features(21, b, h, w))
        bands = S.plan_bands(h, hu, wu, world)
        ex = S.BandExchange(shape, (hu, wu), bands, "cpu", src=0, mode=mode)
        win, row0 = ex.handoff(feat_cpu if rank == 0 else None)
        bd = ex.band
        if rank == 0:                              # the source holds the whole map: crop to its own window too
            win, row0 = win[:, :, bd.a0:bd.a1].contiguous(), bd.a0
        band = D.decode_window(win.to(dev), row0, h, packed, (hu, wu), (bd.y0, bd.y1))
        torch.cuda.synchronize()
        img = ex.gather(band.cpu(), dst=0)
        if rank == 0:
            full = D.decode_features(feat_cpu.to(dev), packed, (hu, wu)).cpu()
            q.put(bool(torch.equal(img, full)))
        else:
            q.put(True)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_decode_bands_on_gpu():
    res = _run(2, "halo", (1, 40, 56, 132, 185), target=_gpu_worker, timeout=600)
    assert all(res)


def _gpu_band_decoder_worker(rank, world, port, mode, geom, q):
    """``BandDecoder`` itself on the device with the host-staged gloo transport (all ranks share cuda:0): side-stream
    hand-off overlapped with rank 0's own band, band-sized windows, two steps over the same buffers, the
    one-message gather -- every line of sharded.py that the RCCL run executes except the wire itself."""
    import diinn_amd.decoder as D
    import diinn_amd.sharded as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        b, h, w, hu, wu = geom
        mode, _, compute = mode.partition(":")                    # "halo:bf16x3" = hand-off mode : arithmetic
        compute = compute or "f32"
        packed = D.pack_state_dict(synth.decoder_state_dict(33)).to(dev)
        dec = S.BandDecoder((b, 64, h, w), (hu, wu), packed, src=0, mode=mode, compute=compute)
        assert dec.host_staged and dec.side is not None
        ok = True
        for it in range(2):
            feat_cpu = torch.from_numpy(synth.encoder_features(33 + it, b, h, w))
            feat = feat_cpu.to(dev) if rank == 0 else None
            band = dec.step(feat)
            img = dec.gather(band, dst=0)
            torch.cuda.synchronize()
            if rank == 0:
                full = D.decode_features(feat, packed, (hu, wu), compute=compute)
                ok = ok and bool(torch.equal(img, full))
            else:
                assert img is None
        q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "halo"), (4, "halo"), (2, "bcast"), (2, "halo:bf16x3"), (2, "halo:bf16_full")])
def test_band_decoder_on_gpu_over_host_staged_gloo(world, mode):
    res = _run(world, mode, (2, 40, 56, 132, 185), target=_gpu_band_decoder_worker, timeout=600)
    assert all(res)


def _gpu_back_to_back_worker(rank, world, port, mode, geom, q):
    """Several steps queued back to back with DIFFERENT features and no synchronisation between them (the
    host-staged transport re-posted its receive into a pinned buffer an asynchronous copy could still be reading, and
    every other test sent the same features each step or synchronised in between): the bands of every step, cloned
    in stream order, must each equal the same rows of that step's unsharded decode."""
    import diinn_amd.decoder as D
    import diinn_amd.sharded as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        b, h, w, hu, wu = geom
        packed = D.pack_state_dict(synth.decoder_state_dict(35)).to(dev)
        dec = S.BandDecoder((b, 64, h, w), (hu, wu), packed, src=0, mode=mode)
        steps = 6
        feats = [torch.from_numpy(synth.encoder_features(100 + it, b, h, w)).to(dev) for it in range(steps)]
        big = torch.randn(2048, 2048, device=dev)
        torch.cuda.synchronize()
        bands, imgs = [], []
        for it in range(steps):
            _ = big @ big                                          # keeps the stream busy: the copies queue up behind it
            band = dec.step(feats[it] if rank == 0 else None)
            bands.append(band.clone())                             # stream-ordered snapshot of the reused output band
            if it % 2 == 1:                                        # and the gather's stages every other step
                img = dec.gather(band, dst=0)
                imgs.append((it, img.clone() if rank == 0 else None))
        torch.cuda.synchronize()
        bd = dec.band
        ok = True
        for it in range(steps):
            full = D.decode_features(feats[it], packed, (hu, wu))
            ok = ok and bool(torch.equal(bands[it], full[:, :, bd.y0:bd.y1]))
        if rank == 0:
            for it, img in imgs:
                ok = ok and bool(torch.equal(img, D.decode_features(feats[it], packed, (hu, wu))))
        q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "halo"), (3, "bcast")])
def test_host_staged_steps_back_to_back_with_changing_features(world, mode):
    res = _run(world, mode, (1, 64, 96, 211, 317), target=_gpu_back_to_back_worker, timeout=600)
    assert all(res)


def _gpu_forward_sharded_toggle_worker(rank, world, port, mode, geom, q):
    """``DIINN.forward_sharded`` caches its band-sized decoder per geometry; the arithmetic and the sine mode are baked
    into it: switching ``decoder.compute`` between two calls must rebuild it, not keep decoding in
    the old arithmetic."""
    import diinn_amd.modules as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        b, h, w, hu, wu = geom
        torch.manual_seed(7)
        model = M.DIINN(mode=3, init_q=False).to(dev).eval()
        x = torch.rand(b, 3, h, w, device=dev)
        outs = {}
        for compute in ("f32", "bf16_full", "f32", "bf16x3"):
            model.decoder.compute = compute
            img = model.forward_sharded(x, (hu, wu), src=0, gather_to=0)
            torch.cuda.synchronize()
            if rank == 0:
                with torch.no_grad():
                    whole = model(x, (hu, wu))
                assert torch.equal(img, whole), f"{compute}: the sharded forward ran another arithmetic"
                outs.setdefault(compute, []).append(img.clone())
        if rank == 0:
            ok = torch.equal(outs["f32"][0], outs["f32"][1]) and not torch.equal(outs["f32"][0], outs["bf16_full"][0])
            model.set_split_bf16()                                   # both optional modes at once: the key sees it too
        else:
            ok = True
            model.set_split_bf16()
        img = model.forward_sharded(x, (hu, wu), src=0, gather_to=0)
        if rank == 0:
            with torch.no_grad():
                ok = ok and bool(torch.equal(img, model(x, (hu, wu))))
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_forward_sharded_follows_a_compute_toggle():
    res = _run(2, "halo", (1, 40, 56, 132, 185), target=_gpu_forward_sharded_toggle_worker, timeout=900)
    assert all(res)


_RCCL_SELF = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
a = torch.randn(2, 64, 66, 256, device=dev); stage = torch.empty_like(a[:, :, 3:40]); win = torch.zeros_like(stage)
big = torch.randn(4096, 4096, device=dev)
side = torch.cuda.Stream(device=dev); cur = torch.cuda.current_stream()
for it in range(3):                               # the buffers are reused, as in a step loop
    a.normal_()
    side.wait_stream(cur)                         # BandExchange.handoff on the source rank, with itself as the peer
    with torch.cuda.stream(side):
        stage.copy_(a[:, :, 3:40])
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, stage, 0), dist.P2POp(dist.irecv, win, 0)])
        for r in reqs:
            r.wait()
    c = big @ big                                 # the rank's own band, queued while the message is on the wire
    cur.wait_stream(side)                         # BandExchange.complete
    torch.cuda.synchronize()
    assert torch.equal(win, a[:, :, 3:40]) and bool(torch.isfinite(c).all())
print("RCCL-SELF-OK")
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_rccl_point_to_point_on_a_side_stream_with_itself_as_peer():
    """No test box has two GPUs, and RCCL refuses two ranks on one.  What CAN run on the real backend is the exact call
    sequence of the overlapped hand-off -- strided stage copy, `batch_isend_irecv` and `wait()` inside a side-stream
    context, the caller's kernels queued meanwhile, `wait_stream` -- with the rank itself as the peer (RCCL supports a
    send and a receive to oneself inside one group)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SELF], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL-SELF-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
