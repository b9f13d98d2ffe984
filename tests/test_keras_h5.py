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


def _bn_permuted(layers, order=(2, 3, 0, 1)):
    """every BatchNormalization layer's four arrays (gamma, beta, moving_mean, moving_variance) listed in another order"""
    out = []
    for ln, ws in layers:
        out.append((ln, [ws[i] for i in order] if ln.startswith("batch_normalization") else ws))
    return out


def test_layer_arrays_are_placed_by_leaf_name(weights):
    layers = _bn_permuted(keras_h5.to_keras_layers(weights))
    assert layers[1][1][0][0].endswith("moving_mean:0")
    back = keras_h5.convert_layers(layers)
    assert all(np.array_equal(back[t.name], weights[t.name]) for t in spec.tensors())
    # a BatchNormalization layer that lists two arrays under the same leaf, or a leaf these layers do not have, is refused by name
    bad = list(keras_h5.to_keras_layers(weights))
    ln, ws = bad[1]
    bad[1] = (ln, [(ws[0][0], ws[0][1]), (ws[1][0], ws[1][1]), (ws[2][0], ws[2][1]), (ln + "/bias:0", ws[3][1])])
    with pytest.raises(ValueError, match=ln):
        keras_h5.convert_layers(bad)
    # unknown names fall back to Keras' own positional rule
    anon = [(ln, [(f"{ln}/w{i}", a) for i, (_, a) in enumerate(ws)]) for ln, ws in keras_h5.to_keras_layers(weights)]
    back = keras_h5.convert_layers(anon)
    assert all(np.array_equal(back[t.name], weights[t.name]) for t in spec.tensors())


def test_mismatch_errors_name_the_layer(weights):
    layers = keras_h5.to_keras_layers(weights)
    ln, ws = layers[7]
    with pytest.raises(ValueError, match=ln):
        keras_h5.convert_layers(layers[:7] + [(ln, ws[:-1])] + layers[8:])
    with pytest.raises(ValueError, match="yaw_new"):
        keras_h5.convert_layers([(n, w[:1]) if n == "yaw_new" else (n, w) for n, w in layers])
    with pytest.raises(ValueError, match="pitch_new"):
        keras_h5.convert_layers([l for l in layers if l[0] != "pitch_new"] + [("dense_7", dict(layers)["pitch_new"])])


@needs_h5py
@pytest.mark.parametrize("variant", ["full_model", "vlen_str", "vlen_bytes", "double_prefix", "bare_names", "bn_permuted", "weightless_layers",
                                     "everything"])
def test_layout_variants(weights, tmp_path, variant):
    """Files this module's own writer did not produce before round 6: a full-model save (model_weights/ + optimizer_weights/), names
    as variable-length str (h5py >= 3) or bytes, weight names with the layer's scope twice or not at all, BatchNormalization arrays in
    another order, layers without weights interleaved.  [The layouts are recollected from Keras 2.1.6 - 2.3; no Keras-written
    file exists offline.]"""
    style = {"double_prefix": "double", "bare_names": "bare", "everything": "double"}.get(variant, "plain")
    layers = keras_h5.to_keras_layers(weights, style=style)
    opts = {}
    if variant in ("full_model", "everything"):
        opts["full_model"] = True
    if variant in ("vlen_str", "everything"):
        opts["names"] = "vlen_str"
    if variant == "vlen_bytes":
        opts["names"] = "vlen_bytes"
    if variant in ("bn_permuted", "everything"):
        layers = _bn_permuted(layers)
    if variant in ("weightless_layers", "everything"):
        opts["weightless"] = [[0, "input_1"], [3, "swish_1"], [40, "add_1"], [133, "global_average_pooling2d_1"]]
    h5 = str(tmp_path / f"{variant}.h5")
    keras_h5.write_keras_h5(h5, layers, **opts)
    blob = keras_h5.load_as_packed(h5)
    assert W.checksum(W.unpack(blob)) == W.checksum(weights)


@needs_h5py
def test_malformed_files_raise_value_error(weights, tmp_path):
    layers = keras_h5.to_keras_layers(weights)
    h5 = str(tmp_path / "short.h5")
    keras_h5.write_keras_h5(h5, layers[:50] + layers[51:])                     # one weighted layer missing
    with pytest.raises(ValueError, match="weighted layers"):
        keras_h5.load_as_packed(h5)
    ln, ws = layers[4]
    keras_h5.write_keras_h5(h5, layers[:4] + [(ln, [(ws[0][0], np.zeros((1, 1, 7, 7), np.float32))])] + layers[5:])
    with pytest.raises(ValueError, match=ln):
        keras_h5.load_as_packed(h5)
