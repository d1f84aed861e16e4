"""CPU: the Python host mirror of the reference interface (no GPU work)."""
import pytest
import torch

import diinn_amd.synth as synth


def test_parameter_names_and_shapes_match_reference_state_dict():
    """SURVEY.md App. A.1 / diinn.py:73-80,92: reference checkpoints must load unchanged."""
    import diinn_amd.decoder as D
    dec = D.ImplicitDecoder(mode=3, init_q=False)
    got = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert got == dict(synth.decoder_param_shapes())
    assert sum(p.numel() for p in dec.parameters()) == 986_627
    missing = dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state_dict().items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


@pytest.mark.parametrize("mode,k1_in", [(1, 256), (2, 832), (3, 832), (4, 832)])
def test_constructor_variants_register_reference_shapes(mode, k1_in):
    import diinn_amd.decoder as D
    dec = D.ImplicitDecoder(mode=mode, init_q=False)
    assert dec.K[1][0].in_channels == k1_in and dec.Q[1][0].in_channels == 256
    assert dec.last_layer.kernel_size == ((3, 3) if mode == 4 else (1, 1))
    dq = D.ImplicitDecoder(mode=mode, init_q=True)
    assert dq.first_layer[0].out_channels == 576 and dq.Q[0][0].in_channels == 576


def test_cpu_tensor_fails_loudly_no_fallback():
    import diinn_amd.decoder as D
    dec = D.ImplicitDecoder(mode=3, init_q=False)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        with torch.no_grad():
            dec(torch.zeros(1, 64, 4, 4), (8, 8))


def test_unsupported_variants_raise():
    import diinn_amd.decoder as D
    for kw in (dict(mode=4), dict(mode=3, init_q=True), dict(mode=1, init_q=True)):
        with pytest.raises(NotImplementedError):
            D.ImplicitDecoder(**kw)(torch.zeros(1, 64, 4, 4), (8, 8))


def test_missing_library_raises(monkeypatch, tmp_path):
    import diinn_amd._native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(N.DiinnNativeError, match="no CPU fallback"):
        N.load()


def test_product_package_never_imports_oracle():
    import os
    from conftest import ROOT
    pkg = os.path.join(ROOT, "dual-interactive-implicit-neural-network_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "diinn_oracle" not in src and "import oracle" not in src, f
