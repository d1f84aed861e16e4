"""CPU: the C-ABI shared library loads, exports every symbol include/diinn_hip.h declares, and its
host-only functions (packing, tables, size queries) are right.  No kernel launches here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import diinn_amd.synth as synth
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "diinn_hip.h")


@pytest.fixture(scope="module")
def lib():
    import diinn_amd.build as b
    b.build()                     # no-op when libdiinn_hip.so is current
    import diinn_amd._native as N
    return N.load()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(diinn_[A-Za-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    import diinn_amd._native as N
    names = declared_symbols()
    assert len(names) >= 17
    raw = C.CDLL(N.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/diinn_hip.h but not exported"
        assert n in N.SIGNATURES, f"{n} has no ctypes signature in _native.py"
    assert sorted(N.SIGNATURES) == names


def test_abi_version_and_status_strings(lib):
    assert lib.diinn_abi_version() == 9
    assert lib.diinn_status_string(0) == b"ok"
    assert b"invalid" in lib.diinn_status_string(1)


def test_debug_knobs_roundtrip(lib):
    """The diagnostic overrides live in one table behind diinn_debug_set / diinn_debug_get (no getenv per launch)."""
    import diinn_amd._native as N
    for name, dflt in [("DIINN_F32_KERNEL", 0), ("DIINN_BF16_KERNEL", 0), ("DIINN_X3_KERNEL", 0), ("DIINN_PBF16_KERNEL", 0), ("DIINN_P_KERNEL", 0),
                       ("DIINN_P_WINO_MIN", 0), ("DIINN_P_X3_MIN", 32768), ("DIINN_ENC_S1_MIN_BLOCKS", 128), ("DIINN_ENC_NO_STREAM1X1", 0),
                       ("DIINN_ENC_LAT_MAX_TILES", 256), ("DIINN_ENC_WINO_MIN", 8192), ("DIINN_ENC_WINO_HALF_MAX", -1),
                       ("DIINN_ENC_WINO_PERSIST", 256), ("DIINN_ENC_X3_MIN", 32768), ("DIINN_ENC_X3_ROWS", 0), ("DIINN_ENC_WINO4_SPLIT", 1)]:
        if name not in os.environ:
            assert N.debug_get(name) == dflt, name
        old = N.debug_get(name)
        N.debug_set(name, 7)
        assert N.debug_get(name) == 7
        N.debug_set(name, old)
    assert lib.diinn_debug_set(b"DIINN_NO_SUCH_KNOB", 1) == 1
    v = C.c_longlong()
    assert lib.diinn_debug_get(b"DIINN_NO_SUCH_KNOB", C.byref(v)) == 1 and lib.diinn_debug_get(None, C.byref(v)) == 1
    # the P algorithm query follows the knob
    a = C.c_int(-1)
    assert lib.diinn_p_launch_info(1, 64, 64, 0, 64, N.COMPUTE_F32, C.byref(a)) == 0 and a.value == N.P_ALGO_WINOGRAD
    assert lib.diinn_p_launch_info(1, 64, 64, 0, 64, N.COMPUTE_BF16_FULL, C.byref(a)) == 0 and a.value == N.P_ALGO_DIRECT_BF16
    N.debug_set("DIINN_P_KERNEL", 1)
    try:
        assert lib.diinn_p_launch_info(1, 64, 64, 0, 64, N.COMPUTE_F32, C.byref(a)) == 0 and a.value == N.P_ALGO_DIRECT
    finally:
        N.debug_set("DIINN_P_KERNEL", 0)
    assert lib.diinn_p_launch_info(1, 64, 64, 5, 5, N.COMPUTE_F32, C.byref(a)) == 1
    assert lib.diinn_p_launch_info(1, 64, 64, 0, 64, 9, C.byref(a)) == 2


def test_host_axis_tables_bit_exact_vs_reference(lib, golden):
    import diinn_amd.decoder as D
    for k in golden.files:
        if not k.startswith("idx/"):
            continue
        _, path, pair = k.split("/")
        n_in, n_out = map(int, pair.split("_"))
        idx, rel = D.axis_tables(n_in, n_out, path == "small")
        assert np.array_equal(idx, golden[k]), k
        assert np.array_equal(rel.view(np.uint32), golden[f"rel/{path}/{pair}"].view(np.uint32)), k


def test_invalid_arguments_return_status_not_crash(lib):
    assert lib.diinn_make_axis_tables(0, 5, 0, None, None) == 1
    assert lib.diinn_workspace_bytes(1, 0, 3) == 0
    r0, r1 = C.c_int(), C.c_int()
    assert lib.diinn_lr_rows_for_band(8, 16, 16, 5, 5, C.byref(r0), C.byref(r1)) == 1
    assert lib.diinn_decode_band(None, None, None, None, 1, 8, 8, 16, 16, 0, 16, 0) == 1
    assert lib.diinn_precompute_P(None, None, None, None, 1, 8, 8, 0, 8) == 1


def test_workspace_and_band_rows(lib):
    import diinn_amd.decoder as D
    assert lib.diinn_workspace_bytes(2, 48, 40) == 2 * 48 * 40 * 1024 * 4
    assert D.lr_rows_for_band(256, 1024, 1024, 0, 1024) == (0, 256)
    assert D.lr_rows_for_band(256, 1024, 1024, 512, 768) == (128, 192)
    import diinn_oracle as orc
    idx, _ = orc.axis_tables(37, 120)
    for (y0, y1) in [(0, 1), (5, 77), (119, 120)]:
        assert D.lr_rows_for_band(37, 120, 171, y0, y1) == (int(idx[y0]), int(idx[y1 - 1]) + 1)


def _chan_of(kk, h):
    m, r = kk >> 4, kk & 15
    return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h


def test_packed_image_layout(lib):
    """Independent numpy restatement of the layout in csrc/diinn_layout.h."""
    import diinn_amd.decoder as D
    sd = synth.decoder_state_dict(11)
    packed = D.pack_state_dict(sd).numpy()
    assert packed.size == lib.diinn_packed_weight_floats() == 986_628 + 196_608 + 393_216 + 294_912 + 768 + 1024 + 393_216 + 1_048_576 + 393_216 + 589_824 + 393_216
    lane = np.arange(64)
    out_l, h_l = lane & 31, lane >> 5
    # WL section
    WL = packed[:3 * 8 * 32 * 2 * 256].reshape(3, 8, 32, 2, 64, 4)
    rng = np.random.default_rng(0)
    for _ in range(200):
        i, m, kg, part, l, e = (int(rng.integers(n)) for n in (3, 8, 32, 2, 64, 4))
        o, cin = 32 * m + (l & 31), _chan_of(4 * kg + e, l >> 5)
        w = sd[f"K.{i + 1}.0.weight"][o, cin, 0, 0] if part == 0 else sd[f"Q.{i + 1}.0.weight"][o, cin, 0, 0]
        assert WL[i, m, kg, part, l, e] == w
    # WP section: [mp][kg][t][lane][e] = Wx[o][c][ky][kx], M-tile mo = 2mp+t
    off = WL.size
    WP = packed[off:off + 32 * 72 * 256].reshape(16, 72, 2, 64, 4)
    for _ in range(200):
        mo, kg, l, e = (int(rng.integers(n)) for n in (32, 72, 64, 4))
        i, ch = mo >> 3, 32 * (mo & 7) + (l & 31)
        kk = 4 * kg + e
        t, c = kk >> 5, 2 * (kk & 31) + (l >> 5)
        col = c * 9 + t + (0 if i == 0 else 256)
        assert WP[mo >> 1, kg, mo & 1, l, e] == sd[f"K.{i}.0.weight"][ch, col, 0, 0]
    off += WP.size
    for i in range(4):
        assert np.array_equal(packed[off + 256 * i: off + 256 * (i + 1)], sd[f"K.{i}.0.bias"])
    off += 1024
    q0 = sd["Q.0.0.weight"][:, :, 0, 0]
    for j in range(3):
        assert np.array_equal(packed[off + 256 * j: off + 256 * (j + 1)], q0[:, j])
    assert np.array_equal(packed[off + 768: off + 1024], sd["Q.0.0.bias"])
    off += 1024
    for i in range(3):
        assert np.array_equal(packed[off + 256 * i: off + 256 * (i + 1)], sd[f"Q.{i + 1}.0.bias"])
    off += 768
    assert np.array_equal(packed[off: off + 768].reshape(3, 256), sd["last_layer.weight"][:, :, 0, 0])
    off += 768
    assert np.array_equal(packed[off: off + 3], sd["last_layer.bias"])
    # WLT section (backward pass): WL with the two channel indices swapped
    WLT = packed[986_628 + 196_608:986_628 + 196_608 + 393_216].reshape(3, 8, 32, 2, 64, 4)
    for _ in range(200):
        i, m, kg, part, l, e = (int(rng.integers(n)) for n in (3, 8, 32, 2, 64, 4))
        cin, o = 32 * m + (l & 31), _chan_of(4 * kg + e, l >> 5)
        w = sd[f"K.{i + 1}.0.weight"][o, cin, 0, 0] if part == 0 else sd[f"Q.{i + 1}.0.weight"][o, cin, 0, 0]
        assert WLT[i, m, kg, part, l, e] == w
    # WPB section (bf16 copy of the 3x3 conv): [mp][ks][t][lane][j], k-step = 4*tap + channel group of 16
    WPB = packed[986_628 + 196_608 + 393_216:986_628 + 196_608 + 393_216 + 294_912].view(np.uint16).reshape(16, 36, 2, 64, 8)
    for _ in range(200):
        mo, ks, l, jj = (int(rng.integers(n)) for n in (32, 36, 64, 8))
        i, ch = mo >> 3, 32 * (mo & 7) + (l & 31)
        tap, c = ks >> 2, 16 * (ks & 3) + 8 * (l >> 5) + jj
        col = c * 9 + tap + (0 if i == 0 else 256)
        w = np.float32(sd[f"K.{i}.0.weight"][ch, col, 0, 0])
        u = int(w.view(np.uint32))
        bf = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF            # round to nearest even
        assert int(WPB[mo >> 1, ks, mo & 1, l, jj]) == bf
    # WLB section (bf16 copy of the per-pixel layers): [layer][m][ks][part][lane][j]; the synthesis rows (part 1) and
    # the BQR table are in revolutions: multiplied by fp32(1/(2 pi)) (before the bf16 rounding for the weights)
    def bf16_bits(w):
        u = int(np.float32(w).view(np.uint32))
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    inv2pi = np.float32(0.15915494309189533577)
    WLB = packed[986_628:986_628 + 196_608].view(np.uint16).reshape(3, 8, 16, 2, 64, 8)
    for _ in range(300):
        i, m, ks, part, l, jj = (int(rng.integers(n)) for n in (3, 8, 16, 2, 64, 8))
        hh = l >> 5
        o, cin = 32 * m + (l & 31), 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (jj >> 2) + 4 * hh + (jj & 3)
        w = sd[f"K.{i + 1}.0.weight"][o, cin, 0, 0] if part == 0 else np.float32(sd[f"Q.{i + 1}.0.weight"][o, cin, 0, 0] * inv2pi)
        assert int(WLB[i, m, ks, part, l, jj]) == bf16_bits(w)
    tail = packed[986_628 + 196_608 + 393_216 + 294_912:]
    BQR = tail[:768].reshape(3, 256)
    for i in range(3):
        assert np.array_equal(BQR[i], (sd[f"Q.{i + 1}.0.bias"] * inv2pi).astype(np.float32))
    Q0R = tail[768:1792].reshape(4, 256)
    for jj in range(3):
        assert np.array_equal(Q0R[jj], (q0[:, jj] * inv2pi).astype(np.float32))
    assert np.array_equal(Q0R[3], (sd["Q.0.0.bias"] * inv2pi).astype(np.float32))
    # WLR: WL with the synthesis pieces (part 1) in revolutions
    WLR = tail[1792:1792 + 393_216].reshape(3, 8, 32, 2, 64, 4)
    assert np.array_equal(WLR[:, :, :, 0], WL[:, :, :, 0])
    assert np.array_equal(WLR[:, :, :, 1], (WL[:, :, :, 1] * inv2pi).astype(np.float32))
    # WPU: the 3x3 conv in Winograd F(2x2,3x3) form, U = G Wx G^T (float64, rounded once), column 2 negated:
    # [mt][row i][sg][col j][lane][e], output 32 mt + (lane & 31), input channel 8 sg + 2 e + (lane >> 5)
    WPU = tail[1792 + 393_216:1792 + 393_216 + 1_048_576].reshape(32, 4, 8, 4, 64, 4)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    for _ in range(300):
        mt, i, sg, jj, l, e = (int(rng.integers(n)) for n in (32, 4, 8, 4, 64, 4))
        layer, ch = mt >> 3, 32 * (mt & 7) + (l & 31)
        c = 8 * sg + 2 * e + (l >> 5)
        col0 = 0 if layer == 0 else 256
        g = sd[f"K.{layer}.0.weight"][ch, col0 + 9 * c: col0 + 9 * c + 9, 0, 0].astype(np.float64).reshape(3, 3)
        u = (G @ g @ G.T)[i, jj]
        assert WPU[mt, i, sg, jj, l, e] == np.float32(-u if jj == 2 else u)
    # WLX (split-bf16 mode): [layer][m][ks][k_hi, q_hi, k_lo, q_lo][lane][j]; hi = the WLB value, lo = bf16(w - hi);
    # hi + lo carries w to 2^-17
    def bf16_val(bits):
        return np.array([bits << 16], dtype=np.uint32).view(np.float32)[0]
    WLX = tail[1792 + 393_216 + 1_048_576:1792 + 393_216 + 1_048_576 + 393_216].view(np.uint16).reshape(3, 8, 16, 4, 64, 8)
    for _ in range(300):
        i, m, ks, part, l, jj = (int(rng.integers(n)) for n in (3, 8, 16, 2, 64, 8))
        hh = l >> 5
        o, cin = 32 * m + (l & 31), 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (jj >> 2) + 4 * hh + (jj & 3)
        w = sd[f"K.{i + 1}.0.weight"][o, cin, 0, 0] if part == 0 else np.float32(sd[f"Q.{i + 1}.0.weight"][o, cin, 0, 0] * inv2pi)
        assert int(WLX[i, m, ks, part, l, jj]) == int(WLB[i, m, ks, part, l, jj])
        hi = bf16_val(int(WLX[i, m, ks, part, l, jj]))
        lo_bits = int(WLX[i, m, ks, 2 + part, l, jj])
        assert lo_bits == bf16_bits(np.float32(w) - hi)
        assert abs(float(hi) + float(bf16_val(lo_bits)) - float(w)) <= 2.0 ** -16 * abs(float(w))
    # WPX (split-bf16 hoisted conv): [og 16][group 4][tap 9][mt 2][hi, lo][lane][j]: output channel 64 og + 32 mt + (lane & 31) of
    # the 1024 (layer og >> 2), input channel 16 group + 8 (lane >> 5) + j
    WPX = tail[1792 + 393_216 + 1_048_576 + 393_216:1792 + 393_216 + 1_048_576 + 393_216 + 589_824].view(np.uint16).reshape(16, 4, 9, 2, 2, 64, 8)
    for _ in range(300):
        og, g, tap, mt, l, jj = (int(rng.integers(n)) for n in (16, 4, 9, 2, 64, 8))
        i, ch = og >> 2, 64 * (og & 3) + 32 * mt + (l & 31)
        c = 16 * g + 8 * (l >> 5) + jj
        col = c * 9 + tap + (0 if i == 0 else 256)
        w = np.float32(sd[f"K.{i}.0.weight"][ch, col, 0, 0])
        hi_bits = bf16_bits(w)
        assert int(WPX[og, g, tap, mt, 0, l, jj]) == hi_bits
        assert int(WPX[og, g, tap, mt, 1, l, jj]) == bf16_bits(w - bf16_val(hi_bits))
    # WL16 (16-pixel fp32 latency kernel, v_mfma_f32_16x16x4_f32 A operands): [layer][wave][i][half][lane][T]: output channel
    # 64 wave + 16 T + (lane & 15), input = the channel at POSITION 4 i + (lane >> 4) of decode_kernel's accumulation order
    WL16 = tail[1792 + 393_216 + 1_048_576 + 393_216 + 589_824:].reshape(3, 4, 64, 2, 64, 4)
    for _ in range(300):
        i, wv, ks, half, l, T = (int(rng.integers(n)) for n in (3, 4, 64, 2, 64, 4))
        pos = 4 * ks + (l >> 4)
        o, cin = 64 * wv + 16 * T + (l & 15), _chan_of(pos >> 1, pos & 1)
        w = sd[f"K.{i + 1}.0.weight"][o, cin, 0, 0] if half == 0 else np.float32(sd[f"Q.{i + 1}.0.weight"][o, cin, 0, 0] * inv2pi)
        assert WL16[i, wv, ks, half, l, T] == w
    # every channel appears exactly once per lane-half in the activation register order
    seen = sorted(_chan_of(kk, h) for kk in range(128) for h in range(2))
    assert seen == list(range(256))


def test_host_axis_tables_match_oracle_on_random_pairs(lib):
    """The C-ABI coordinate code (csrc/diinn_layout.h axis_eval) against the numpy oracle on random
    size pairs, both ATen index-kernel variants: bit-exact."""
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    rng = np.random.default_rng(2024)
    for _ in range(400):
        n_in = int(rng.integers(1, 1500))
        n_out = int(rng.integers(1, 6000))
        for small in (False, True):
            idx, rel = D.axis_tables(n_in, n_out, small)
            oi, orl = orc.axis_tables(n_in, n_out, small)
            assert np.array_equal(idx, oi), (n_in, n_out, small)
            assert np.array_equal(rel.view(np.uint32), orl.view(np.uint32)), (n_in, n_out, small)


def test_window_rows_and_window_validation(lib):
    """diinn_window_rows (host) against the oracle's tables, and the window entry points refuse windows that
    do not hold what the band reads -- checked before any launch, so this runs without a GPU."""
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    import diinn_oracle as orc
    for (h, hu, wu, y0, y1) in [(256, 1024, 1024, 512, 640), (720, 2376, 4224, 0, 297), (1024, 8192, 8192, 7168, 8192),
                                (5, 3, 9, 2, 3)]:
        idx, _ = orc.axis_tables(h, hu, orc.uses_small_output_kernel(hu, wu))
        (a0, an), (r0, rn) = D.window_rows(h, hu, wu, y0, y1)
        assert r0 == idx[y0] and r0 + rn == idx[y1 - 1] + 1
        assert a0 == max(r0 - 1, 0) and a0 + an == min(r0 + rn + 1, h)
    fake = C.c_void_p(4096)          # never dereferenced: validation fails first
    # P window [10,14) cannot serve HR rows 0..8 of a x4 decode (cells 0..1)
    assert lib.diinn_decode_band_win(None, fake, 10, 4, fake, fake, 0, 8, 1, 64, 64, 256, 256, 0, 8, 2, 0) == N.ERR_INVALID_ARG
    # output window must contain the band
    assert lib.diinn_decode_band_win(None, fake, 0, 2, fake, fake, 4, 8, 1, 64, 64, 256, 256, 0, 8, 2, 0) == N.ERR_INVALID_ARG
    # feature window without the halo row
    assert lib.diinn_precompute_P_win(None, fake, 8, 8, fake, fake, 8, 8, 1, 64, 64, 8, 16, 0) == N.ERR_INVALID_ARG
    # window outside the map
    assert lib.diinn_precompute_P_win(None, fake, 60, 8, fake, fake, 61, 3, 1, 64, 64, 61, 64, 0) == N.ERR_INVALID_ARG


def test_tile_entry_point_validates_ranges_and_strides(lib):
    """diinn_decode_tile_win (ABI v7): bad column ranges and strides that would make rows / planes / batch items
    interleave are refused before anything is launched (host-side checks only: no GPU needed for these calls)."""
    import ctypes as C
    dummy = C.c_void_p(16)                                       # never dereferenced: every call below fails validation

    def tile(y0, y1, x0, x1, rs, ps, bs, p_row0=0, p_rows=64, compute=0, sin=2):
        return lib.diinn_decode_tile_win(None, dummy, p_row0, p_rows, dummy, dummy, rs, ps, bs, 1, 64, 64, 256, 256,
                                         y0, y1, x0, x1, sin, compute)
    ok_strides = (40, 40 * 8, 3 * 40 * 8)
    assert tile(0, 8, 10, 10, *ok_strides) == 1                 # empty column range
    assert tile(0, 8, -1, 10, *ok_strides) == 1
    assert tile(0, 8, 250, 257, *ok_strides) == 1               # past the image
    assert tile(8, 8, 0, 16, *ok_strides) == 1                  # empty row range
    assert tile(0, 8, 0, 41, *ok_strides) == 1                  # row stride shorter than the tile's width
    assert tile(0, 8, 0, 40, 40, 40 * 7, 3 * 40 * 8) == 1       # planes would overlap
    assert tile(0, 8, 0, 40, 40, 40 * 8, 2 * 40 * 8) == 1       # batch items would overlap
    assert tile(0, 8, 0, 40, *ok_strides, p_row0=5, p_rows=4) == 1   # the P window does not hold the tile's LR rows
    assert tile(0, 8, 0, 40, *ok_strides, compute=99) == 2      # unsupported arithmetic
    assert tile(0, 8, 0, 40, *ok_strides, sin=7) == 2


def test_conv_wino4_entry_point_validates_its_arguments(lib):
    """diinn_conv_wino4 / diinn_rdn_forward_wino4 (Winograd F(4x4,3x3) encoder layers): null pointers, channel counts that are
    not multiples of 8, unaligned images and empty maps are refused before anything is launched (no GPU needed)."""
    import ctypes as C
    d = C.c_void_p(4096)                                         # never dereferenced: every call below fails validation

    def conv(inp=d, cin=64, packed=d, bias=d, out=d, b=1, h=64, w=64):
        return lib.diinn_conv_wino4(None, inp, cin * h * w, cin, packed, bias, None, 0, out, 64 * h * w, 1, b, h, w)
    assert conv(inp=None) == 1 and conv(packed=None) == 1 and conv(bias=None) == 1 and conv(out=None) == 1
    assert conv(cin=12) == 2 and conv(cin=0) == 2
    assert conv(packed=C.c_void_p(4100)) == 1                    # the weight image is read with 16-byte loads
    assert conv(b=0) == 1 and conv(h=0) == 1
    assert lib.diinn_rdn_forward_wino4(None, d, d, d, None, d, d, d, 1, 64, 64) == 1     # without the F(4x4) image
    # the F(2x2) image may be NULL only where the map takes F(4x4): 128 x 128 runs F(2x2), so a NULL there is refused (before
    # anything is launched)
    assert lib.diinn_rdn_wino4_applies(1, 128, 128) == 0
    assert lib.diinn_rdn_forward_wino4(None, d, d, None, d, d, d, d, 1, 128, 128) == 1
    assert lib.diinn_rdn_wino4_packed_floats() == lib.diinn_rdn_wino_packed_floats() // 16 * 36
    # the workspace form: too small / misaligned workspaces (validated before the launch)
    wsf = lib.diinn_conv_wino4_workspace_floats()
    assert wsf == 1024 + 2 * (256 + 8) * 16384 and lib.diinn_rdn_workspace_floats(1, 8, 8) == wsf + 64 * 2240

    def conv_ws(ws=d, floats=wsf, cin=64):
        return lib.diinn_conv_wino4_ws(None, d, cin * 64 * 64, cin, d, d, None, 0, d, 64 * 64 * 64, 1, 1, 64, 64, ws, floats)
    assert conv_ws(floats=wsf - 1) == 1 and conv_ws(ws=C.c_void_p(4100)) == 1 and conv_ws(cin=12) == 2
    # the ONE trunk entry point (ABI v9): algo caps the kernel family; the images that family reads must be given, others
    # may be NULL; planes / split area must be 16-byte aligned; the split area may be NULL (nothing is split then)
    A = {"auto": 0, "direct": 1, "wino": 2, "wino4": 3, "x3": 4}

    def ex(algo, packed=d, wino=d, wino4=d, x3=d, planes=d, area=d, sfe1=d, biases=d, out=d, b=1, h=64, w=64):
        return lib.diinn_rdn_forward_ex(None, A[algo], sfe1, packed, wino, wino4, x3, biases, planes, area, out, b, h, w)
    for algo in A:
        assert ex(algo, sfe1=None) == 1 and ex(algo, packed=None) == 1 and ex(algo, biases=None) == 1 and ex(algo, out=None) == 1
        assert ex(algo, planes=None) == 1 and ex(algo, planes=C.c_void_p(4100)) == 1 and ex(algo, area=C.c_void_p(4100)) == 1
        assert ex(algo, b=0) == 1 and ex(algo, h=0) == 1
    assert ex("wino", wino=None) == 1 and ex("wino4", wino4=None) == 1 and ex("x3", x3=None) == 1 and ex("x3", wino=None) == 1
    assert ex("wino4", wino=None, h=128, w=128) == 1             # 128 x 128 runs F(2x2): its image is needed (as the v8 wrapper)
    assert lib.diinn_rdn_forward_ex(None, 5, d, d, d, d, d, d, d, d, d, 1, 64, 64) == 1 and \
        lib.diinn_rdn_forward_ex(None, -1, d, d, d, d, d, d, d, d, d, 1, 64, 64) == 1
    assert lib.diinn_rdn_planes_floats(A["wino4"], 1, 8, 8) == 64 * 2240 and lib.diinn_rdn_planes_floats(A["x3"], 2, 8, 8) == 128 * 2816
    assert lib.diinn_rdn_planes_floats(7, 1, 8, 8) == 0 and lib.diinn_rdn_planes_floats(0, 0, 8, 8) == 0
    assert lib.diinn_rdn_x3_workspace_floats(1, 8, 8) == wsf + 64 * 2816   # the deprecated wrappers' single workspace
    st = C.c_int(0)
    assert lib.diinn_conv_wino4_ws_status(None, None, 0, C.byref(st)) == 1 and lib.diinn_conv_wino4_ws_status(None, d, 0, None) == 1
    info = (C.c_int * 4)()
    assert lib.diinn_decode_kernel_info(1, 96, 96, 0, 96, 0, 96, 0, info) == 0 and info[0] in (1, 3)
    assert lib.diinn_decode_kernel_info(1, 96, 96, 0, 97, 0, 96, 0, info) == 1 and lib.diinn_decode_kernel_info(1, 96, 96, 0, 96, 0, 96, 9, info) == 2
    assert lib.diinn_decode_kernel_info(1, 1024, 1024, 0, 1024, 0, 1024, 0, info) == 0 and list(info) == [1, 64, 128, 1]
