"""Training path: gradients of the mode-3 decoder.

CPU part: the oracle's autograd against the REAL reference's .grad fixtures
(tests/golden/diinn_golden_grad.npz), and the product's backward formulas
(``training.backward_from_saved``) fed with oracle-computed saved planes against the same fixtures.
GPU part: the full autograd function (HIP forward with saved activations + library-GEMM backward)
against fixtures and the float64 oracle."""
import os

import numpy as np
import pytest
import torch

import diinn_amd.synth as synth
import diinn_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROW_STRIDE = 8


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "diinn_golden_grad.npz"))


def grad_cases(gold):
    for key in gold.files:
        if key.startswith("meta/"):
            name = key[5:]
            b, h, w, hu, wu, gain = gold[key]
            yield name, int(b), int(h), int(w), int(hu), int(wu), float(gain)


def _inputs(name, b, h, w, hu, wu, gain):
    sd = synth.decoder_state_dict(123, gain)
    feat = synth.encoder_features(123, b, h, w)
    r = synth.uniform(123, f"gradw:{name}", (b, 3, hu, wu), 1.0)
    return sd, feat, r


def _check_against_fixture(gold, name, d_feat, grads, rtol):
    """max|g - ref| <= rtol * max|ref| per tensor (K weights: every 8th output row is pinned)."""
    ref = gold[f"grad/{name}/feat"]
    err = float(np.abs(d_feat - ref).max())
    assert err <= rtol * float(np.abs(ref).max()), f"{name} d_feat err {err:.3e}"
    for pname, g in grads.items():
        ref = gold[f"grad/{name}/{pname}"]
        if pname.startswith("K.") and pname.endswith("weight"):
            g = g[::ROW_STRIDE]
        assert g.shape == ref.shape, (pname, g.shape, ref.shape)
        err = float(np.abs(g - ref).max())
        assert err <= rtol * max(float(np.abs(ref).max()), 1e-6), f"{name} {pname} err {err:.3e}"


def test_oracle_autograd_matches_reference_fixture(gold):
    for name, b, h, w, hu, wu, gain in grad_cases(gold):
        sd, feat, r = _inputs(name, b, h, w, hu, wu, gain)
        out, d_feat, grads = orc.reference_gradients(sd, feat, (hu, wu), r)
        assert float(np.abs(out.numpy() - gold[f"out/{name}"]).max()) <= 1e-6 * max(1.0, float(np.abs(gold[f"out/{name}"]).max()))
        _check_against_fixture(gold, name, d_feat.numpy(), {k: v.numpy() for k, v in grads.items()}, 2e-5)


def test_pack_gather_index_is_the_host_packer():
    """The device re-pack (one gather) reproduces diinn_pack_weights outside the bf16 section."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_amd.training as T
    lib = N.load()
    sd = synth.decoder_state_dict(7)
    idx = T.pack_gather_index()
    flat = torch.cat([torch.from_numpy(sd[n]).reshape(-1) for n in T.PARAM_NAMES] + [torch.zeros(1)])
    got = flat[idx].numpy()
    ref = D.pack_state_dict(sd).numpy()
    off, size = C.c_size_t(), C.c_size_t()
    keep = np.ones(ref.size, bool)
    for section in (7, 9, 10, 11, 12, 13, 14, 15, 16):     # inference-only sections: derived values, left zero by the gather
        assert lib.diinn_packed_section(section, C.byref(off), C.byref(size)) == 0
        keep[off.value:off.value + size.value] = False
    assert lib.diinn_packed_section(6, C.byref(off), C.byref(size)) == 0
    word = off.value + 3                           # the validity word: the magic in a host-packed image, 0 in a gathered one
    assert ref[word:word + 1].view(np.uint32)[0] == N.PACKED_MAGIC and got[word] == 0.0
    keep[word] = False
    assert np.array_equal(got[keep], ref[keep]) and not got[~keep].any()
    total = 0
    for s in range(17):
        o, z = C.c_size_t(), C.c_size_t()
        assert lib.diinn_packed_section(s, C.byref(o), C.byref(z)) == 0
        assert o.value == total                    # sections are contiguous
        total = o.value + z.value
    assert total == ref.size == lib.diinn_packed_weight_floats()
    assert lib.diinn_packed_section(17, C.byref(off), C.byref(size)) != 0


def test_backward_formulas_on_cpu(gold):
    """training.backward_from_saved (the product's backward algebra) with oracle-computed saved planes."""
    import diinn_amd.training as T
    for name, b, h, w, hu, wu, gain in grad_cases(gold):
        sd, feat, r = _inputs(name, b, h, w, hu, wu, gain)
        _, acts = orc.saved_planes(sd, feat, (hu, wu))
        params = [torch.from_numpy(sd[n]) for n in T.PARAM_NAMES]
        d_feat, d_params = T.backward_from_saved(torch.from_numpy(r), torch.from_numpy(feat), acts, params, (hu, wu))
        grads = {n: g.numpy() for n, g in zip(T.PARAM_NAMES, d_params)}
        for n in T.PARAM_NAMES:
            assert grads[n].shape == tuple(T.PARAM_SHAPES[n])
        _check_against_fixture(gold, name, d_feat.numpy(), grads, 5e-5)


def _train_forward(lib, N, sd_packed, feat, b, h, w, hu, wu, dev, fill=float("nan")):
    """precompute_P + decode_kernel<SAVE> through the C ABI; returns (out, tiled acts [4,T,512,32])."""
    import ctypes as C
    n = b * hu * wu
    t = (n + 31) // 32
    assert lib.diinn_training_plane_floats(n, 512) == t * 512 * 32
    ws = torch.empty(b * h * w * 1024, device=dev)
    acts = torch.full((4, t, 512, 32), fill, device=dev)
    out = torch.empty((b, 3, hu, wu), device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N.check(lib.diinn_precompute_P(stream, C.c_void_p(feat.data_ptr()), C.c_void_p(sd_packed.data_ptr()),
                                   C.c_void_p(ws.data_ptr()), b, h, w, 0, h), "P")
    N.check(lib.diinn_decode_train_fwd(stream, C.c_void_p(ws.data_ptr()), C.c_void_p(sd_packed.data_ptr()),
                                       C.c_void_p(out.data_ptr()), C.c_void_p(acts.data_ptr()),
                                       b, h, w, hu, wu, N.SIN_DEFAULT), "train_fwd")
    torch.cuda.synchronize()
    return out, acts


@pytest.mark.gpu
def test_hip_training_forward_saves_the_oracle_planes(gold):
    """decode_kernel<SAVE>: output equals the inference kernel's, saved planes equal the oracle's."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_amd.training as T
    dev = torch.device("cuda:0")
    lib = N.load()
    for (b, h, w, hu, wu, gain) in [(2, 12, 10, 31, 27, 1.0), (1, 9, 14, 36, 56, 3.0), (1, 20, 33, 47, 130, 1.0)]:
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref_out, ref_acts = orc.saved_planes(sd, feat, (hu, wu))
        packed = D.pack_state_dict(sd).to(dev)
        f = torch.from_numpy(feat).to(dev)
        n = b * hu * wu
        out, acts = _train_forward(lib, N, packed, f, b, h, w, hu, wu, dev)
        infer = D.decode_features(f, packed, (hu, wu))
        # the inference kernel keeps the synthesis branch in revolutions (weights / (2 pi), sine = v_sin(x - rint(x))) and
        # takes P from the Winograd form of the hoisted conv; the training forward works in radians (it saves the sine
        # arguments) on the direct conv: same mathematics, fp32 rounding apart (the x3 stress weights amplify it)
        assert float((out - infer).abs().max()) <= 2e-5 * gain * max(1.0, float(ref_out.abs().max()))
        tol = 1e-4 * max(1.0, float(ref_out.abs().max()))
        assert float((out.cpu() - ref_out).abs().max()) <= tol
        flat = acts.permute(0, 2, 1, 3).reshape(4, 512, -1)
        assert torch.isnan(flat[..., n:]).all()            # padding of the last tile is never written
        a = T.untile_planes(acts, n).view(4, 2, 256, n).cpu()
        assert torch.isfinite(a).all()                     # every plane element written
        scale = max(1.0, float(ref_acts.abs().max()))
        assert float((a - ref_acts).abs().max()) <= 2e-5 * scale
    assert lib.diinn_training_plane_floats(1 << 31, 512) == -1


@pytest.mark.gpu
def test_autograd_through_hip_decoder(gold):
    """ImplicitDecoder.forward(x, size) with autograd on (training call, sr_module.py:128): gradients of
    every parameter and of the features against the real reference's fixtures and the fp64 oracle."""
    import diinn_amd.decoder as D
    dev = torch.device("cuda:0")
    for name, b, h, w, hu, wu, gain in grad_cases(gold):
        sd, feat, r = _inputs(name, b, h, w, hu, wu, gain)
        dec = D.ImplicitDecoder(mode=3, init_q=False)
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        dec = dec.to(dev).train()
        x = torch.from_numpy(feat).to(dev).requires_grad_(True)
        y = dec(x, [hu, wu])
        (y * torch.from_numpy(r).to(dev)).sum().backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.cpu().numpy() for n, p in dec.named_parameters()}
        _check_against_fixture(gold, name, x.grad.cpu().numpy(), grads, 1e-4)
        # full tensors (not only the pinned rows) against the float64 oracle
        _, d_feat64, g64 = orc.reference_gradients(sd, feat, (hu, wu), r, dtype=torch.float64)
        for n, g in grads.items():
            ref = g64[n].numpy()
            assert float(np.abs(g - ref).max()) <= 1e-4 * max(float(np.abs(ref).max()), 1e-6), n
        assert float(np.abs(x.grad.cpu().numpy() - d_feat64.numpy()).max()) <= 1e-4 * float(d_feat64.abs().max())


@pytest.mark.gpu
def test_training_step_decreases_loss():
    """A few Adam steps of SRLitModule.step (sr_module.py:113-125,127-129) on one synthetic batch: the
    L1 loss goes down, i.e. encoder and decoder gradients flow through the HIP forward."""
    import diinn_amd.modules as M
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = M.SRLitModule(arch="diinn", mode=3, init_q=False).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    lr = torch.rand(2, 3, 16, 16, device=dev)
    batch = {2: (lr, torch.rand(2, 3, 32, 32, device=dev), ["a", "b"]),
             3: (lr, torch.rand(2, 3, 48, 48, device=dev), ["a", "b"])}
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss, _ = net.step(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_fused_backward_equals_formula_backward_on_gpu():
    """bwd_head_kernel + bwd_layer_kernel (diinn_backward_data), plane_gemm / rowdot / cell_sum kernels
    against the same gradients stated as plain tensor algebra (backward_from_saved) on the same saved
    planes, at sizes with a ragged last tile and rectified-to-zero channels; also the planes themselves."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_amd.training as T
    dev = torch.device("cuda:0")
    lib = N.load()
    for (b, h, w, hu, wu, gain) in [(3, 17, 13, 50, 41, 1.0), (1, 8, 8, 24, 24, 3.0), (2, 5, 40, 3, 9, 1.0)]:
        sd = synth.decoder_state_dict(5, gain)
        feat = torch.from_numpy(synth.encoder_features(5, b, h, w)).to(dev)
        params = [torch.from_numpy(sd[n]).to(dev) for n in T.PARAM_NAMES]
        gout = torch.from_numpy(synth.uniform(5, "g", (b, 3, hu, wu), 1.0)).to(dev)
        n = b * hu * wu
        t = (n + 31) // 32
        packed = T.pack_on_device(params)
        # the permutation sections agree with the host packer; the pad word behind bL is the validity word of the
        # derived sections (the magic on the host, 0 in a gathered image)
        host = D.pack_state_dict(sd)
        assert torch.equal(packed[: 986_627].cpu(), host[: 986_627])
        # ... "DIWP" on the device image, whose WPU section (13: the hoisted conv in Winograd form, filled on the device in
        # float64 in the host packer's operation order) is bit-identical to the host packer's; every other derived section zero
        assert packed[986_627:986_628].view(torch.int32).item() == N.PACKED_MAGIC_WPU
        o13, z13 = T._section(13)
        assert torch.equal(packed[o13:o13 + z13].cpu(), host[o13:o13 + z13])
        for sec in (7, 9, 10, 11, 12, 14, 15, 16):
            o, z = T._section(sec)
            assert not bool(packed[o:o + z].any())
        # P from the training image on the Winograd kernel == P from the inference image, bit for bit; an inference decode
        # from the training image is refused (NaN): only diinn_precompute_P_wpu accepts the WPU-only word
        ptr_ = lambda x: C.c_void_p(x.data_ptr())                 # noqa: E731
        st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p_a, p_b = torch.empty((b, h, w, 1024), device=dev), torch.empty((b, h, w, 1024), device=dev)
        hostd = host.to(dev)
        N.check(lib.diinn_precompute_P_wpu(st0, ptr_(feat), ptr_(packed), ptr_(p_a), b, h, w, 0, h), "P_wpu")
        N.check(lib.diinn_precompute_P_ex(st0, ptr_(feat), ptr_(hostd), ptr_(p_b), b, h, w, 0, h, 0), "P_ex")
        assert torch.equal(p_a, p_b)
        N.check(lib.diinn_precompute_P_ex(st0, ptr_(feat), ptr_(packed), ptr_(p_b), b, h, w, 0, h, 0), "P_ex")
        assert bool(torch.isnan(p_b).all())
        blank = packed.clone()
        blank[986_627] = 0.0
        N.check(lib.diinn_precompute_P_wpu(st0, ptr_(feat), ptr_(blank), ptr_(p_b), b, h, w, 0, h), "P_wpu")
        assert bool(torch.isnan(p_b).all())
        out, acts_t = _train_forward(lib, N, packed, feat, b, h, w, hu, wu, dev, fill=0.0)
        acts = T.untile_planes(acts_t, n).view(4, 2, 256, n)
        # planes
        gp = gout.permute(1, 0, 2, 3).reshape(3, n).contiguous()
        g_t = torch.full((4, t, 512, 32), float("nan"), device=dev)
        q_t = torch.full((4, t, 256, 32), float("nan"), device=dev)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(lib.diinn_backward_data(stream, C.c_void_p(gp.data_ptr()), C.c_void_p(acts_t.data_ptr()),
                                        C.c_void_p(packed.data_ptr()), C.c_void_p(g_t.data_ptr()),
                                        C.c_void_p(q_t.data_ptr()), n), "bwd")
        torch.cuda.synchronize()
        # round 6: the head's gates are computed inside layer 3's kernel; bwd_head_kernel as a launch of its own
        # (DIINN_TRAIN_SPLIT_HEAD = 1) leaves bit-identical planes, padding included
        g_s = torch.full((4, t, 512, 32), float("nan"), device=dev)
        q_s = torch.full((4, t, 256, 32), float("nan"), device=dev)
        N.debug_set("DIINN_TRAIN_SPLIT_HEAD", 1)
        try:
            N.check(lib.diinn_backward_data(stream, C.c_void_p(gp.data_ptr()), C.c_void_p(acts_t.data_ptr()),
                                            C.c_void_p(packed.data_ptr()), C.c_void_p(g_s.data_ptr()),
                                            C.c_void_p(q_s.data_ptr()), n), "bwd")
        finally:
            N.debug_set("DIINN_TRAIN_SPLIT_HEAD", 0)
        torch.cuda.synchronize()
        same = lambda x, y: bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())   # noqa: E731
        assert same(g_t, g_s) and same(q_t, q_s), (b, h, w)
        g = T.untile_planes(g_t, n).view(4, 2, 256, n)
        q = T.untile_planes(q_t, n)
        assert torch.isfinite(g).all() and torch.isfinite(q).all()
        if n % 32:
            assert torch.isnan(g_t.permute(0, 2, 1, 3).reshape(4, 512, -1)[..., n:]).all()   # padding untouched
        a64 = acts.double()
        q_ref = a64[:, 0] * torch.sin(a64[:, 1])
        assert float((q.double() - q_ref).abs().max()) <= 2e-6 * max(1.0, float(q_ref.abs().max()))
        g_q = params[T.PARAM_NAMES.index("last_layer.weight")].view(3, 256).double().t() @ gp.double()
        for i in (3, 2, 1, 0):
            ga = g_q * torch.sin(a64[i, 1]) * (a64[i, 0] > 0)
            gs = g_q * a64[i, 0] * torch.cos(a64[i, 1])
            scale = max(float(ga.abs().max()), float(gs.abs().max()), 1e-12)
            assert float((g[i, 0].double() - ga).abs().max()) <= 2e-5 * scale, i
            assert float((g[i, 1].double() - gs).abs().max()) <= 2e-5 * scale, i
            if i:
                wq = params[T.PARAM_NAMES.index(f"K.{i}.0.weight")].view(256, 832)[:, :256].double()
                qw = params[T.PARAM_NAMES.index(f"Q.{i}.0.weight")].view(256, 256).double()
                g_q = wq.t() @ ga + qw.t() @ gs
        # gradients
        df_a, dp_a = T.backward_fused(gout, feat, acts_t, params, packed, (hu, wu))
        df_b, dp_b = T.backward_from_saved(gout, feat, acts, params, (hu, wu))
        torch.cuda.synchronize()
        assert float((df_a - df_b).abs().max()) <= 5e-5 * float(df_b.abs().max())
        for name, x, y in zip(T.PARAM_NAMES, dp_a, dp_b):
            assert x.shape == y.shape
            assert float((x - y).abs().max()) <= 5e-5 * max(float(y.abs().max()), 1e-6), name
    assert lib.diinn_backward_data(None, C.c_void_p(gp.data_ptr()), C.c_void_p(acts_t.data_ptr()),
                                   C.c_void_p(packed.data_ptr()), C.c_void_p(g_t.data_ptr()),
                                   C.c_void_p(q_t.data_ptr()), 0) == N.ERR_INVALID_ARG


@pytest.mark.gpu
def test_cell_sum_and_unfold_kernels():
    """cell_sum_kernel (round 6: one 16- / 8-byte load per cell row at the integer scales x4 / x2, scalar loads elsewhere) against
    the one-hot-GEMM statement of the same sums (training._cell_sum) -- x4, x2, x3, non-integer, down-scaling, Wu % 4 != 0, batches
    -- with its optional tiled copy (the A operand of the hoisted conv's weight-gradient GEMM) equal to tile_planes of the NCHW
    result; unfold_tiled_kernel equal to the reference's F.unfold(feat, 3, padding=1), tiled, rows 576.. zero."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.training as T
    dev = torch.device("cuda:0")
    lib = N.load()
    gen = torch.Generator(device=dev).manual_seed(11)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda x: C.c_void_p(x.data_ptr())                      # noqa: E731
    for (b, h, w, hu, wu) in [(2, 12, 12, 48, 48), (1, 9, 10, 18, 20), (2, 8, 8, 24, 24), (1, 7, 5, 23, 18), (1, 16, 12, 8, 6),
                              (3, 6, 6, 24, 26), (1, 48, 48, 192, 192)]:
        n = b * hu * wu
        t = (n + 31) // 32
        g_t = torch.randn((4, t, 512, 32), device=dev, generator=gen)
        geo = T._geometry(b, h, w, hu, wu, dev)
        idx_h, _, idx_w, _, _ = T.coordinate_tensors(h, w, hu, wu, dev)
        dp = torch.full((b, 1024, h, w), float("nan"), device=dev)
        cells = b * h * w
        tc = (cells + 31) // 32
        dp_t = torch.zeros((tc, 1024, 32), device=dev)
        N.check(lib.diinn_backward_cell_sum_ex(stream, ptr(g_t), ptr(geo["seg_h"]), ptr(geo["seg_w"]), ptr(dp), ptr(dp_t), b, h, w, hu, wu), "cell_sum")
        dp2 = torch.full((b, 1024, h, w), float("nan"), device=dev)
        N.check(lib.diinn_backward_cell_sum(stream, ptr(g_t), ptr(geo["seg_h"]), ptr(geo["seg_w"]), ptr(dp2), b, h, w, hu, wu), "cell_sum")
        torch.cuda.synchronize()
        assert torch.equal(dp, dp2)
        ga = T.untile_planes(g_t, n)[:, :256].reshape(1024, n)                       # g_a rows of the four layers
        ref = T._cell_sum(ga.double(), b, hu, wu, h, w, idx_h, idx_w)
        assert float((dp.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (b, h, w, hu, wu)
        assert torch.equal(dp_t, T.tile_planes(dp.permute(1, 0, 2, 3).reshape(1024, cells)))
        feat = torch.randn((b, 64, h, w), device=dev, generator=gen)
        u_t = torch.full((tc, 640, 32), float("nan"), device=dev)
        N.check(lib.diinn_unfold_tiled(stream, ptr(feat), ptr(u_t), 640, b, h, w), "unfold")
        unf = torch.nn.functional.unfold(feat, 3, padding=1).permute(1, 0, 2).reshape(576, cells)
        assert torch.equal(u_t, T.tile_planes(torch.cat([unf, unf.new_zeros((64, cells))], 0)))
    assert lib.diinn_unfold_tiled(stream, ptr(feat), ptr(u_t), 100, b, h, w) == N.ERR_INVALID_ARG
    assert lib.diinn_unfold_tiled(stream, None, ptr(u_t), 640, b, h, w) == N.ERR_INVALID_ARG
    assert lib.diinn_backward_cell_sum_ex(stream, ptr(g_t), ptr(geo["seg_h"]), ptr(geo["seg_w"]), None, None, b, h, w, hu, wu) == N.ERR_INVALID_ARG


@pytest.mark.gpu
def test_plane_gemm_kernel():
    """C = A . B^T over the pixel axis of tiled planes with split-K partials and the row-sum column, and the
    skinny rowdot product, against float64, for ragged pixel counts, row windows and empty trailing splits."""
    import ctypes as C
    import diinn_amd._native as N
    import diinn_amd.training as T
    dev = torch.device("cuda:0")
    lib = N.load()
    gen = torch.Generator(device=dev).manual_seed(1)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda x: C.c_void_p(x.data_ptr())                      # noqa: E731
    for (a_rows, a0, m, b_rows, b0, nc, npix, ksplit) in [(128, 0, 128, 256, 0, 256, 1000, 3), (512, 0, 512, 256, 0, 256, 4099, 64),
                                                          (512, 256, 256, 768, 256, 512, 77, 8), (128, 0, 128, 256, 0, 256, 31, 1)]:
        a_full = torch.randn((a_rows, npix), device=dev, generator=gen)
        b_full = torch.randn((b_rows, npix), device=dev, generator=gen)
        a_t, b_t = T.tile_planes(a_full), T.tile_planes(b_full)
        if npix % 32:                                             # garbage in the padding must not matter
            a_t.view(-1, a_rows, 32)[-1, :, npix % 32:] = float("nan")
            b_t.view(-1, b_rows, 32)[-1, :, npix % 32:] = float("inf")
        part = torch.full((ksplit, m, nc + 1), float("nan"), device=dev)
        N.check(lib.diinn_plane_gemm_nt(stream, ptr(a_t), a_rows, a0, ptr(b_t), b_rows, b0, ptr(part), m, nc, npix, ksplit, 1),
                "plane_gemm")
        torch.cuda.synchronize()
        got = part.double().sum(0)
        a, b = a_full[a0:a0 + m].double(), b_full[b0:b0 + nc].double()
        ref = torch.cat([a @ b.t(), a.sum(1, keepdim=True)], dim=1)
        assert torch.isfinite(got).all()
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * max(1.0, (npix / 1000) ** 0.5), (m, nc, npix)
        part2 = torch.full((ksplit, m, nc), float("nan"), device=dev)
        N.check(lib.diinn_plane_gemm_nt(stream, ptr(a_t), a_rows, a0, ptr(b_t), b_rows, b0, ptr(part2), m, nc, npix, ksplit, 0),
                "plane_gemm")
        torch.cuda.synchronize()
        assert torch.equal(part2, part[:, :, :nc])
        # skinny product against a 4-row group
        s_full = torch.randn((4, npix), device=dev, generator=gen)
        s_t = T.tile_planes(s_full)
        splits = 5
        pr = torch.full((splits, a_rows, 4), float("nan"), device=dev)
        N.check(lib.diinn_plane_rowdot(stream, ptr(a_t), a_rows, ptr(s_t), ptr(pr), a_rows, npix, splits), "rowdot")
        torch.cuda.synchronize()
        ref = a_full.double() @ s_full.double().t()
        assert float((pr.double().sum(0) - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) * max(1.0, (npix / 1000) ** 0.5)
    assert lib.diinn_plane_gemm_nt(None, ptr(a_t), 128, 0, ptr(b_t), 256, 0, ptr(part), 256, 256, 31, 1, 0) == N.ERR_INVALID_ARG
    assert lib.diinn_plane_gemm_nt(None, ptr(a_t), 128, 0, ptr(b_t), 256, 0, ptr(part), 64, 256, 31, 1, 0) == N.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_hoisted_conv_gradients_on_the_library_kernels():
    """training._conv_grads_native: the hoisted 3x3 convolution's input gradient as a 1024 -> 64 convolution on the encoder's
    kernels (split-K / Winograd F(2x2) / F(4x4) by map size) and its weight gradient as unfold + plane GEMM, against
    torch.nn.grad (MIOpen): batch 1 (a permuted view that needs its copy), ragged maps (the untiled fallback), the training
    geometry.  The F(4x4) input gradient at B = 16, 48x48 differs by 2.5e-5 of its maximum (F(2x2) / split-K: 1e-6)."""
    import diinn_amd.training as T
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (b, h, w) in [(1, 8, 8), (2, 8, 8), (1, 6, 5), (2, 12, 16), (3, 7, 9), (1, 96, 100), (16, 48, 48)]:
        feat = torch.randn(b, 64, h, w, device=dev)
        wx = torch.randn(1024, 64, 3, 3, device=dev) * 0.05
        dp = torch.randn(b, 1024, h, w, device=dev)
        dw_ref = torch.nn.grad.conv2d_weight(feat, wx.shape, dp, padding=1).reshape(1024, 576)
        df_ref = torch.nn.grad.conv2d_input(feat.shape, wx, dp, padding=1)
        dw, df = T._conv_grads_native(feat, wx, dp, True)
        torch.cuda.synchronize()
        assert float((dw - dw_ref).abs().max()) <= 1e-5 * float(dw_ref.abs().max()), (b, h, w)
        assert float((df - df_ref).abs().max()) <= 5e-5 * float(df_ref.abs().max()), (b, h, w)
        dw2, df2 = T._conv_grads_native(feat, wx, dp, True, want_weight=False)
        assert dw2 is None and torch.equal(df2, df)
