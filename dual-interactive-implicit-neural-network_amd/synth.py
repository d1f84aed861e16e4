"""Deterministic synthetic inputs for the DIINN decode path.

Counter-based generator keyed by ``(seed, tensor_name, flat_index)`` so that
the golden-fixture script (which runs next to the reference), the CPU oracle
tests and the GPU parity tests all regenerate *bit-identical* fp32 weights and
encoder features without shipping megabytes of tensors.  Only integer hashing
and exactly-representable float arithmetic are used (no libm calls), so the
values do not depend on the host CPU or the numpy build.

Distributions follow SURVEY.md §8(d2):
  * weights / biases ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in))  -- what
    ``ImplicitDecoder(mode=3)`` produces with PyTorch's default Conv2d init
    (reference: src/models/components/diinn.py:73-80,92);
  * encoder features ~ approximately N(0,1) (Irwin-Hall, 12 uniforms - 6).

Parameter names and shapes are the reference's state_dict entries
(SURVEY.md App. A.1).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

IN_CHANNELS = 64
HIDDEN = 256
N_LAYERS = 4
UNFOLD = IN_CHANNELS * 9  # 576


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _name_key(seed: int, name: str) -> np.uint64:
    h = 0xCBF29CE484222325  # FNV-1a 64
    for ch in name.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    h ^= (seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return _mix(np.array([h], dtype=np.uint64))[0]


def _bits24(seed: int, name: str, n: int, stream: int = 0) -> np.ndarray:
    """n integers in [0, 2^24), one per flat index."""
    key = _name_key(seed, f"{name}#{stream}")
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * _GOLDEN + key
    return (_mix(idx) >> np.uint64(40)).astype(np.int64)


def uniform(seed: int, name: str, shape, bound: float) -> np.ndarray:
    """fp32 array ~ U(-bound, bound); value = fp32((2u-1)*bound), u = k/2^24."""
    n = int(np.prod(shape))
    k = _bits24(seed, name, n)
    u = (2.0 * k.astype(np.float64) + 1.0) / float(1 << 24) - 1.0  # exact in f64
    return (u * float(bound)).astype(np.float32).reshape(shape)


def normalish(seed: int, name: str, shape) -> np.ndarray:
    """fp32 array, approx N(0,1): sum of 12 uniforms - 6 (exact f64 arithmetic)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for s in range(12):
        acc += _bits24(seed, name, n, stream=s + 1).astype(np.float64)
    acc = acc / float(1 << 24) - 6.0
    return acc.astype(np.float32).reshape(shape)


def decoder_param_shapes(mode: int = 3) -> "OrderedDict[str, tuple]":
    """Reference ``ImplicitDecoder(mode=mode, init_q=False).state_dict()`` layout
    (diinn.py:53-92).  Mode 1 chains k -> K[i] (256 inputs); modes 2-4 feed
    [k or q ; unfolded features] (832 inputs)."""
    shapes: "OrderedDict[str, tuple]" = OrderedDict()
    for i in range(N_LAYERS):
        kin = UNFOLD if i == 0 else (HIDDEN if mode == 1 else HIDDEN + UNFOLD)
        qin = 3 if i == 0 else HIDDEN
        shapes[f"K.{i}.0.weight"] = (HIDDEN, kin, 1, 1)
        shapes[f"K.{i}.0.bias"] = (HIDDEN,)
        shapes[f"Q.{i}.0.weight"] = (HIDDEN, qin, 1, 1)
        shapes[f"Q.{i}.0.bias"] = (HIDDEN,)
    shapes["last_layer.weight"] = (3, HIDDEN, 1, 1)
    shapes["last_layer.bias"] = (3,)
    return shapes


SIREN_Q_GAIN = (30.0, math.sqrt(6.0))


def decoder_state_dict(seed: int = 123, gain: float = 1.0, mode: int = 3,
                       q_gain=None) -> "OrderedDict[str, np.ndarray]":
    """Synthetic decoder weights in the reference's state_dict naming.

    ``gain`` scales every tensor (gain=3 is the SURVEY §8(d2) stress set:
    larger sine arguments and output magnitude).  ``q_gain = (first, hidden)`` additionally scales the
    synthesis branch's WEIGHTS (``Q.0.0.weight`` by ``first``, ``Q.1..3.0.weight`` by ``hidden``; biases and
    every other tensor untouched): ``SIREN_Q_GAIN`` = (30, sqrt 6) is the trained-SIREN range -- layer-0 sine
    arguments of tens of radians on the raw coordinates (reference diinn.py:61-62,134) with |out| still O(1)."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    shapes = decoder_param_shapes(mode)
    for name, shape in shapes.items():
        layer = name.rsplit(".", 1)[0] + ".weight"
        wshape = shapes[layer]
        fan_in = wshape[1] * wshape[2] * wshape[3]
        bound = 1.0 / math.sqrt(fan_in)
        sd[name] = (uniform(seed, name, shape, bound) * np.float32(gain)).astype(np.float32)
        if q_gain is not None and name.startswith("Q.") and name.endswith(".weight"):
            sd[name] = (sd[name] * np.float32(q_gain[0] if name.startswith("Q.0.") else q_gain[1])).astype(np.float32)
    return sd


def encoder_features(seed: int, b: int, h: int, w: int, tag: str = "feat") -> np.ndarray:
    """Synthetic LR encoder feature map [B,64,H,W] fp32 (stands in for RDN output)."""
    return normalish(seed, f"{tag}:{b}x{h}x{w}", (b, IN_CHANNELS, h, w))


def layer_gain(gain_seed: int, layer: str) -> float:
    """A per-LAYER gain 2^u, u ~ U(-0.6, 1.0) (0.66 .. 2.0), keyed by (gain_seed, layer name): a trained network has gains of its
    own in every layer, a uniformly scaled default init has not (round 6 fixtures: tests/golden/make_golden_r6.py)."""
    u = float(uniform(gain_seed, "layer_gain:" + layer, (1,), 0.8)[0]) + 0.2
    return float(2.0 ** u)


def state_dict_for(shapes, seed: int = 123, prefix: str = "", gain: float = 1.0,
                   layer_gain_seed=None) -> "OrderedDict[str, np.ndarray]":
    """Synthetic tensors for any conv-style module given ``{name: shape}`` (names ending in
    ``.weight``/``.bias``): U(+-gain/sqrt(fan_in)) with fan_in taken from the sibling weight; ``layer_gain_seed``: every
    layer's weight and bias additionally scaled by ``layer_gain(layer_gain_seed, layer)``."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        shape = tuple(int(x) for x in shape)
        layer = name.rsplit(".", 1)[0]
        wname = layer + ".weight"
        wshape = tuple(int(x) for x in shapes.get(wname, shape))
        fan_in = int(np.prod(wshape[1:])) if len(wshape) > 1 else 1
        g = gain * (layer_gain(layer_gain_seed, prefix + layer) if layer_gain_seed is not None else 1.0)
        out[name] = uniform(seed, prefix + name, shape, g / math.sqrt(max(fan_in, 1)))
    return out
