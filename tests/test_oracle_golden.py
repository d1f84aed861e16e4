"""CPU: the oracle (oracle/diinn_oracle.py) against the fixtures captured from the REAL
reference decoder (tests/golden/make_golden.py imports /root/reference in the build container).
This is what pins the oracle; the GPU tests then compare the HIP path with it."""
import numpy as np
import torch

import diinn_amd.synth as synth
import diinn_oracle as orc
from conftest import golden_cases


def test_axis_tables_bit_exact(golden):
    """idx/rel tables of _make_pos_encoding + nearest-exact (diinn.py:94-110,168), both ATen CPU kernels."""
    n = 0
    for k in golden.files:
        if not k.startswith("idx/"):
            continue
        _, path, pair = k.split("/")
        n_in, n_out = map(int, pair.split("_"))
        idx, rel = orc.axis_tables(n_in, n_out, path == "small")
        assert np.array_equal(idx, golden[k]), k
        assert np.array_equal(rel.view(np.uint32), golden[f"rel/{path}/{pair}"].view(np.uint32)), k
        n += 1
    assert n >= 40


def test_ratio_constant(golden):
    for k in golden.files:
        if k.startswith("ratio/"):
            h, w, hu, wu = map(int, k[6:].split("_"))
            assert np.float32(golden[k][0]) == orc.scale_ratio(h, w, hu, wu)


def test_small_output_rule():
    assert orc.uses_small_output_kernel(64, 64) and not orc.uses_small_output_kernel(64, 65)


def test_reference_form_matches_reference_outputs(golden):
    """decode_reference_form == ImplicitDecoder.forward outputs (same ATen conv kernels: bit-exact here;
    the assertion allows 1e-6 for other hosts)."""
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden[f"out/{name}"]
        for bsize in (None, 30000):
            got = orc.decode_reference_form(sd, feat, (hu, wu), bsize).numpy()
            assert got.shape == ref.shape
            assert float(np.abs(got - ref).max()) <= 1e-6 * max(1.0, float(np.abs(ref).max())), (name, bsize)


def test_reference_form_matches_siren_range_outputs(golden_r4):
    """The same on the SIREN-range fixtures (Q.0 x 30, Q.1-3 x sqrt 6: layer-0 sine arguments of ~33-67 rad), and the
    hoisted form the kernels use stays at rounding noise there too."""
    from conftest import siren_cases
    for name, b, h, w, hu, wu, gain, qg in siren_cases(golden_r4):
        sd = synth.decoder_state_dict(123, gain, q_gain=qg)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden_r4[f"out/{name}"]
        scale = max(1.0, float(np.abs(ref).max()))
        got = orc.decode_reference_form(sd, feat, (hu, wu), 30000).numpy()
        assert float(np.abs(got - ref).max()) <= 1e-6 * scale, name
        hoisted = orc.decode_hoisted_form(sd, feat, (hu, wu)).numpy()
        assert float(np.abs(hoisted - ref).max()) <= 2e-5 * scale, name


def test_float64_differences_are_reference_rounding_noise(golden, golden_r4):
    """d64 = ref64 - ref32 of every fixture: ~2e-8 .. 9e-8 at default init, 8e-6 on the x3 stress set."""
    from conftest import golden_cases
    for name, *_rest, gain in golden_cases(golden):
        d = float(np.abs(golden_r4[f"d64/{name}"]).max())
        assert d <= (1e-7 if gain == 1.0 else 2e-5), (name, d)
        assert golden_r4[f"d64/{name}"].shape == golden[f"out/{name}"].shape


def test_hoisted_form_is_within_rounding(golden):
    """The per-cell hoist the kernels use equals the reference to rounding noise."""
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden[f"out/{name}"]
        got = orc.decode_hoisted_form(sd, feat, (hu, wu)).numpy()
        assert float(np.abs(got - ref).max()) <= 5e-6 * max(1.0, float(np.abs(ref).max())), name


def test_row_band_equals_slice_of_full():
    sd = synth.decoder_state_dict(3)
    feat = synth.encoder_features(3, 1, 20, 24)
    full = orc.decode_reference_form(sd, feat, (66, 80), None)
    band = orc.decode_reference_form(sd, feat, (66, 80), None, row_range=(17, 41))
    assert float((full[:, :, 17:41] - band).abs().max()) <= 1e-6


def test_unfold_matches_torch_unfold():
    x = torch.from_numpy(synth.encoder_features(1, 2, 7, 5))
    ours = orc.unfold3x3(x)
    theirs = torch.nn.functional.unfold(x, 3, padding=1).view(2, 64 * 9, 7, 5)
    assert torch.equal(ours, theirs)


def test_synth_is_deterministic_and_shaped():
    sd = synth.decoder_state_dict(123)
    assert list(sd) == list(synth.decoder_param_shapes())
    assert sum(v.size for v in sd.values()) == 986_627
    a = synth.encoder_features(123, 1, 8, 8)
    b = synth.encoder_features(123, 1, 8, 8)
    assert np.array_equal(a, b) and abs(float(a.mean())) < 0.1 and 0.8 < float(a.std()) < 1.2
    assert float(np.abs(sd["K.0.0.weight"]).max()) <= 1 / 24.0


def test_plain_c_oracle_tables_bit_exact(golden):
    """Third, independent restatement (plain C, oracle/diinn_oracle_c.c) of the coordinate code."""
    import diinn_oracle_c as oc
    for k in golden.files:
        if not k.startswith("idx/"):
            continue
        _, path, pair = k.split("/")
        n_in, n_out = map(int, pair.split("_"))
        idx, rel = oc.axis_tables(n_in, n_out, path == "small")
        assert np.array_equal(idx, golden[k]), k
        assert np.array_equal(rel.view(np.uint32), golden[f"rel/{path}/{pair}"].view(np.uint32)), k


def test_plain_c_oracle_matches_reference_outputs(golden):
    """The per-pixel C evaluation of diinn.py:163-173 as written (no hoisting, index-order sums)
    against the real reference's outputs; small fixtures only (it is a scalar loop)."""
    import diinn_oracle_c as oc
    for name, b, h, w, hu, wu, gain in golden_cases(golden):
        if b * hu * wu > 12000:
            continue
        sd = synth.decoder_state_dict(123, gain)
        feat = synth.encoder_features(123, b, h, w)
        ref = golden[f"out/{name}"]
        got = oc.decode(sd, feat, (hu, wu))
        err = float(np.abs(got - ref).max())
        assert err <= 5e-6 * max(1.0, float(np.abs(ref).max())), (name, err)   # index-order fp32 sums vs mkldnn blocking


def test_oracle_modes_1_and_2_match_reference(golden):
    """Ablation modes of the reference decoder (diinn.py:116-131) restated in the oracle."""
    feat = synth.encoder_features(123, 1, 24, 20)
    for mode in (1, 2):
        sd = synth.decoder_state_dict(123, mode=mode)
        ref = golden[f"mode{mode}/out_24x20_79x66"]
        got = orc.decode_reference_form(sd, feat, (79, 66), 30000, mode=mode).numpy()
        assert float(np.abs(got - ref).max()) <= 1e-6
