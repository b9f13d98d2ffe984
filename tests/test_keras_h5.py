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


@needs_h5py
def test_h5_roundtrip_through_hdf5(weights, tmp_path):
    h5 = str(tmp_path / "WHENet.h5")
    keras_h5.write_keras_h5(h5, keras_h5.to_keras_layers(weights, offset=82))   # as if built second in a session
    blob = keras_h5.load_as_packed(h5)
    assert W.checksum(W.unpack(blob)) == W.checksum(weights)
    assert os.path.exists(h5 + ".whnp")                         # cached for the next construction
    assert keras_h5.load_as_packed(h5) == blob
    with pytest.raises(ValueError):
        p = tmp_path / "junk.h5"
        p.write_bytes(b"not hdf5 at all")
        keras_h5.load_as_packed(str(p))
