"""LIIF comparison decoder: oracle and C-ABI tables against fixtures captured from the
real reference (tests/golden/make_golden_liif.py); on the GPU, liif_kernel against fixtures and oracle."""
import json
import os

import numpy as np
import pytest
import torch

import diinn_amd.synth as synth
import liif_oracle as L

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "liif_golden.npz"))


def _cases(gold):
    for k in gold.files:
        if k.startswith("meta/"):
            b, h, w, hu, wu, gain = gold[k]
            yield k[5:], int(b), int(h), int(w), int(hu), int(wu), float(gain)


def _imnet(gold, gain):
    shapes = {k: v for k, v in json.loads(str(gold["liif/shapes_json"])).items() if k.startswith("imnet.")}
    return synth.state_dict_for(shapes, 123, "liif.", gain=gain)


def test_oracle_tables_and_outputs_match_reference(gold):
    n = 0
    for k in gold.files:
        if k.startswith("idx/"):
            n_in, n_out, v = map(int, k[4:].split("_"))
            idx, rel = L.liif_axis_tables(n_in, n_out, v)
            assert np.array_equal(idx, gold[k]), k
            assert np.array_equal(rel.view(np.uint32), gold["rel/" + k[4:]].view(np.uint32)), k
            n += 1
    assert n >= 20
    for name, b, h, w, hu, wu, gain in _cases(gold):
        out = L.liif_query_reference_form(_imnet(gold, gain), synth.encoder_features(123, b, h, w), (hu, wu)).numpy()
        ref = gold[f"out/{name}"]
        assert float(np.abs(out - ref).max()) <= 1e-6 * max(1.0, float(np.abs(ref).max())), name


def test_host_liif_tables_are_bit_exact(gold):
    """C ABI diinn_liif_make_axis_tables (csrc/diinn_layout.h liif_axis_eval, shared with the kernel)."""
    import diinn_amd.decoder as D
    for k in gold.files:
        if k.startswith("idx/"):
            n_in, n_out, v = map(int, k[4:].split("_"))
            idx, rel, cell = D.liif_axis_tables(n_in, n_out, v)
            assert np.array_equal(idx, gold[k]), k
            assert np.array_equal(rel.view(np.uint32), gold["rel/" + k[4:]].view(np.uint32)), k
            assert np.float32(cell) == L.liif_rel_cell(n_in, n_out)
    rng = np.random.default_rng(7)
    for _ in range(200):
        n_in, n_out = int(rng.integers(1, 700)), int(rng.integers(1, 3000))
        for v in (-1, 1):
            idx, rel, _ = D.liif_axis_tables(n_in, n_out, v)
            oi, orl = L.liif_axis_tables(n_in, n_out, v)
            assert np.array_equal(idx, oi) and np.array_equal(rel.view(np.uint32), orl.view(np.uint32)), (n_in, n_out, v)


def test_liif_module_has_reference_parameter_names(gold):
    import diinn_amd.modules as M
    net = M.make_net("liif", 3, False)
    ref = json.loads(str(gold["liif/shapes_json"]))
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == ref


@pytest.mark.gpu
def test_liif_kernel_matches_reference_fixtures(gold):
    import diinn_amd.decoder as D
    dev = torch.device("cuda:0")
    for name, b, h, w, hu, wu, gain in _cases(gold):
        sd = _imnet(gold, gain)
        packed = D.pack_liif_state_dict(sd).to(dev)
        feat = torch.from_numpy(synth.encoder_features(123, b, h, w)).to(dev)
        out = D.liif_decode_features(feat, packed, (hu, wu))
        torch.cuda.synchronize()
        ref = gold[f"out/{name}"]
        err = float(np.abs(out.cpu().numpy() - ref).max())
        assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), f"{name}: {err:.3e}"


@pytest.mark.gpu
def test_liif_model_end_to_end_and_larger_shape(gold):
    """Full LIIF (RDN encoder on PyTorch-ROCm + HIP decoder) against the reference's forward; and a 256x256 x4
    decode against the oracle on a row band."""
    import diinn_amd.decoder as D
    import diinn_amd.modules as M
    dev = torch.device("cuda:0")
    full = json.loads(str(gold["liif/shapes_json"]))
    net = M.LIIF()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(full, 123, "liifnet.").items()})
    net = net.to(dev).eval()
    img = torch.from_numpy(synth.uniform(123, "img:1x3x12x10", (1, 3, 12, 10), 0.5) + np.float32(0.5)).to(dev)
    with torch.no_grad():
        y = net(img, [31, 27], 300)
    ref = gold["liif/out_1x3x12x10_to_31x27"]
    assert float(np.abs(y.cpu().numpy() - ref).max()) <= 2e-4 * max(1.0, float(np.abs(ref).max()))   # encoder on MIOpen vs CPU
    with pytest.raises(NotImplementedError):
        net(img, [31, 27])                                   # grad enabled: inference only
    sd = _imnet(gold, 1.0)
    feat = synth.encoder_features(5, 1, 96, 80)
    out = D.liif_decode_features(torch.from_numpy(feat).to(dev), D.pack_liif_state_dict(sd).to(dev), (384, 301))
    torch.cuda.synchronize()
    ref = L.liif_query_reference_form(sd, feat, (384, 301)).numpy()
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= 1e-4
