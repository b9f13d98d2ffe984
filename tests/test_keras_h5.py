"""Keras HDF5 -> WHNPACK1 conversion.  WHENet.h5 itself is absent from
the reference, so the file under test is written here in the Keras-2.1.6 layout (auto-numbered
layer names, weight_names attrs) from the synthetic snapshot, by the h5py helper interpreter."""
import os

import numpy as np
import pytest

from whenet_hip import keras_h5, spec, weights as W

needs_h5py = pytest.mark.skipif(not os.path.exists(keras_h5.HELPER), reason="no interpreter with h5py")


def test_keras_layer_census(weights):
    layers = keras_h5.to_keras_layers(weights)
    assert len(layers) == 133                                   # 81 conv/dw + 49 BN + 3 Dense
    assert sum(len(ws) for _, ws in layers) == 315
    names = [ln for ln, _ in layers]
    assert names[0] == "conv2d_1" and names[1] == "batch_normalization_1" and names[2] == "depthwise_conv2d_1"
    assert names[-3:] == ["yaw_new", "pitch_new", "roll_new"]
    assert sum(n.startswith("conv2d_") for n in names) == 65    # 33 Conv2D + 32 SE convs
    back = keras_h5.convert_layers(layers)
    assert all(np.array_equal(back[t.name], weights[t.name]) for t in spec.tensors())


def test_positional_pairing_rejects_mismatches(weights):
    layers = keras_h5.to_keras_layers(weights)
    with pytest.raises(ValueError):
        keras_h5.convert_layers(layers[:-4] + layers[-3:])      # one weighted layer missing
    bad = list(layers)
    ln, ws = bad[5]
    bad[5] = (ln, [(ws[0][0], np.zeros((3, 3, 1, 1), np.float32))] + ws[1:])
    with pytest.raises(ValueError):
        keras_h5.convert_layers(bad)
    with pytest.raises(ValueError):
        keras_h5.convert_layers([(("dense_1" if n == "yaw_new" else n), w) for n, w in layers])


def shuffled_heads(layers, order=("roll_new", "yaw_new", "pitch_new")):
    """The same file with the three Dense heads stored in another order (same-depth layers of the
    Keras graph may be serialised in any order): name matching, not position, must place them."""
    body = [l for l in layers if l[0] not in order]
    by = dict(layers)
    return body + [(n, by[n]) for n in order]


def test_heads_are_matched_by_name_not_position(weights):
    layers = shuffled_heads(keras_h5.to_keras_layers(weights))
    assert [ln for ln, _ in layers][-3:] == ["roll_new", "yaw_new", "pitch_new"]
    back = keras_h5.convert_layers(layers)
    assert all(np.array_equal(back[t.name], weights[t.name]) for t in spec.tensors())


@needs_h5py
def test_h5_roundtrip_through_hdf5(weights, tmp_path, monkeypatch):
    h5 = str(tmp_path / "WHENet.h5")
    keras_h5.write_keras_h5(h5, keras_h5.to_keras_layers(weights, offset=82))   # as if built second in a session
    blob = keras_h5.load_as_packed(h5)
    assert W.checksum(W.unpack(blob)) == W.checksum(weights)
    assert not os.path.exists(h5 + ".whnp")                     # caching is opt-in
    # opt-in cache, keyed on the CONTENT of the .h5: a replaced file is re-converted
    # whatever its mtime, a stray .whnp without a matching digest is ignored
    monkeypatch.setenv("WHENET_H5_CACHE", "1")
    assert keras_h5.load_as_packed(h5) == blob and os.path.exists(h5 + ".whnp") and os.path.exists(h5 + ".whnp.sha256")
    other = dict(weights)
    other["yaw/bias"] = other["yaw/bias"] + 1.0
    st = os.stat(h5)
    keras_h5.write_keras_h5(h5, keras_h5.to_keras_layers(other))
    os.utime(h5, (st.st_atime, st.st_mtime - 1000))             # older than the cache, as `cp -p` would leave it
    blob2 = keras_h5.load_as_packed(h5)
    assert W.checksum(W.unpack(blob2)) == W.checksum(other) != W.checksum(weights)
    with open(h5 + ".whnp", "wb") as f:                         # poisoned cache: digest no longer matches the blob,
        f.write(blob)                                           # but it matches the file -> detectable only by key
    with open(h5 + ".whnp.sha256", "w") as f:
        f.write("0" * 64)
    assert keras_h5.load_as_packed(h5) == blob2
    with pytest.raises(ValueError):
        p = tmp_path / "junk.h5"
        p.write_bytes(b"not hdf5 at all")
        keras_h5.load_as_packed(str(p))
