"""MetaSR comparison decoder: oracle and C-ABI tables against fixtures captured from the
real reference (tests/golden/make_golden_metasr.py); on the GPU, metasr_kernel against fixtures and oracle."""
import json
import os

import numpy as np
import pytest
import torch

import diinn_amd.synth as synth
import metasr_oracle as MO

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "metasr_golden.npz"))


def _cases(gold):
    for k in gold.files:
        if k.startswith("meta/"):
            b, h, w, hu, wu, gain = gold[k]
            yield k[5:], int(b), int(h), int(w), int(hu), int(wu), float(gain)


def _imnet(gold, gain):
    shapes = {k: v for k, v in json.loads(str(gold["metasr/shapes_json"])).items() if k.startswith("imnet.")}
    return synth.state_dict_for(shapes, 123, "metasr.", gain=gain)


def test_oracle_tables_and_outputs_match_reference(gold):
    for k in gold.files:
        if k.startswith("idx/"):
            n_in, n_out = map(int, k[4:].split("_"))
            idx, rel = MO.metasr_axis_tables(n_in, n_out)
            assert np.array_equal(idx, gold[k]), k
            assert np.array_equal(rel.view(np.uint32), gold["rel/" + k[4:]].view(np.uint32)), k
    for name, b, h, w, hu, wu, gain in _cases(gold):
        out = MO.metasr_query_reference_form(_imnet(gold, gain), synth.encoder_features(123, b, h, w), (hu, wu)).numpy()
        ref = gold[f"out/{name}"]
        assert float(np.abs(out - ref).max()) <= 1e-6 * max(1.0, float(np.abs(ref).max())), name


def test_host_tables_and_packing(gold):
    import diinn_amd._native as N
    import diinn_amd.decoder as D
    for k in gold.files:
        if k.startswith("idx/"):
            n_in, n_out = map(int, k[4:].split("_"))
            idx, rel, r_rev = D.metasr_axis_tables(n_in, n_out)
            assert np.array_equal(idx, gold[k]), k
            assert np.array_equal(rel.view(np.uint32), gold["rel/" + k[4:]].view(np.uint32)), k
            assert np.float32(r_rev) == MO.metasr_r_rev(n_in, n_out)
    rng = np.random.default_rng(3)
    for _ in range(200):
        n_in, n_out = int(rng.integers(1, 700)), int(rng.integers(1, 3000))
        idx, rel, _ = D.metasr_axis_tables(n_in, n_out)
        oi, orl = MO.metasr_axis_tables(n_in, n_out)
        assert np.array_equal(idx, oi) and np.array_equal(rel.view(np.uint32), orl.view(np.uint32)), (n_in, n_out)
    # packed image: independent restatement of the layout in csrc/diinn_layout.h
    sd = _imnet(gold, 1.0)
    packed = D.pack_metasr_state_dict(sd).numpy()
    assert packed.size == N.load().diinn_metasr_packed_floats() == 3 * 18 * 32 * 256 + 1024 + 1728
    w2 = packed[:3 * 18 * 32 * 256].reshape(3, 18, 32, 64, 4)
    for _ in range(300):
        o, mm, kg, l, e = (int(rng.integers(n)) for n in (3, 18, 32, 64, 4))
        kk = 4 * kg + e
        cin = 32 * (kk >> 4) + (kk & 3) + 8 * ((kk & 15) >> 2) + 4 * (l >> 5)
        assert w2[o, mm, kg, l, e] == sd["imnet.layers.2.weight"][3 * (32 * mm + (l & 31)) + o, cin]
    off = w2.size
    assert np.array_equal(packed[off:off + 768].reshape(3, 256), sd["imnet.layers.0.weight"].T)
    assert np.array_equal(packed[off + 768:off + 1024], sd["imnet.layers.0.bias"])
    assert np.array_equal(packed[off + 1024:].reshape(3, 576), sd["imnet.layers.2.bias"].reshape(576, 3).T)


def test_metasr_module_has_reference_parameter_names(gold):
    import diinn_amd.modules as M
    net = M.make_net("metasr", 3, False)
    ref = json.loads(str(gold["metasr/shapes_json"]))
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == ref


@pytest.mark.gpu
def test_metasr_kernel_matches_reference_fixtures(gold):
    import diinn_amd.decoder as D
    dev = torch.device("cuda:0")
    for name, b, h, w, hu, wu, gain in _cases(gold):
        sd = _imnet(gold, gain)
        packed = D.pack_metasr_state_dict(sd).to(dev)
        feat = torch.from_numpy(synth.encoder_features(123, b, h, w)).to(dev)
        out = D.metasr_decode_features(feat, packed, (hu, wu))
        torch.cuda.synchronize()
        ref = gold[f"out/{name}"]
        err = float(np.abs(out.cpu().numpy() - ref).max())
        assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), f"{name}: {err:.3e}"


@pytest.mark.gpu
def test_metasr_model_end_to_end_and_larger_shape(gold):
    import diinn_amd.decoder as D
    import diinn_amd.modules as M
    dev = torch.device("cuda:0")
    full = json.loads(str(gold["metasr/shapes_json"]))
    net = M.MetaSR()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(full, 123, "metasrnet.").items()})
    net = net.to(dev).eval()
    img = torch.from_numpy(synth.uniform(123, "img:1x3x12x10", (1, 3, 12, 10), 0.5) + np.float32(0.5)).to(dev)
    with torch.no_grad():
        y = net(img, [31, 27], 300)
    ref = gold["metasr/out_1x3x12x10_to_31x27"]
    assert float(np.abs(y.cpu().numpy() - ref).max()) <= 2e-4 * max(1.0, float(np.abs(ref).max()))
    sd = _imnet(gold, 1.0)
    feat = synth.encoder_features(5, 1, 96, 80)
    out = D.metasr_decode_features(torch.from_numpy(feat).to(dev), D.pack_metasr_state_dict(sd).to(dev), (384, 301))
    torch.cuda.synchronize()
    ref = MO.metasr_query_reference_form(sd, feat, (384, 301)).numpy()
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
