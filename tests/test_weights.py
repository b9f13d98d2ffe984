"""Packed snapshot format + the seeded synthetic snapshot."""
import numpy as np
import pytest

from whenet_hip import spec, weights as W


def test_pack_roundtrip(weights):
    blob = W.pack(weights)
    back = W.unpack(blob)
    assert set(back) == {t.name for t in spec.tensors()}
    for t in spec.tensors():
        assert back[t.name].shape == t.shape
        assert np.array_equal(back[t.name], weights[t.name])


def test_pack_rejects_bad_shape(weights):
    w = dict(weights)
    w["yaw/bias"] = np.zeros(7, np.float32)
    with pytest.raises(ValueError):
        W.pack(w)
    w = dict(weights)
    del w["stem/conv/kernel"]
    with pytest.raises(ValueError):
        W.pack(w)


def test_unpack_rejects_garbage():
    with pytest.raises(ValueError):
        W.unpack(b"not a snapshot at all, definitely")


def test_synthetic_is_pinned(weights, golden):
    # the golden vectors were generated on exactly these bytes
    assert W.checksum(weights) == golden["info"]["weights_sha256"]


def test_synthetic_sane(weights):
    for bn in spec.bn_names():
        assert np.all(weights[f"{bn}/var"] > 0)
    assert np.isfinite(np.concatenate([v.ravel() for v in weights.values()])).all()
