"""The oracle has no reference-side vectors to be pinned against ("parity unpinned");
these tests pin it the only ways available here:
  * committed self-generated goldens (regression),
  * an independent float32 torch/oneDNN restatement agrees to float32 noise,
  * an independent third-party implementation of EfficientNet-B0 (HuggingFace
    transformers, torch) fed the same weights produces the same 7x7x1280 features,
  * the reference's own pre/post-processing lines restated literally.
"""
import numpy as np
import pytest

from oracle import whenet_oracle as O
from oracle.whenet_torch import TorchWHENet
from whenet_hip import spec


def test_golden_regression(weights, golden):
    crops = golden["crops"][:3]
    r = O.forward(crops, weights, np.float64)
    np.testing.assert_allclose(r["logits"], golden["expected"]["logits"][:3], rtol=0, atol=1e-9)
    ang = np.stack([r["yaw"], r["pitch"], r["roll"]], axis=1)
    np.testing.assert_allclose(ang, golden["expected"]["angles"][:3], rtol=0, atol=1e-9)
    assert np.array_equal(r["argmax"], golden["expected"]["argmax"][:3])


def test_torch_restatement_agrees(weights, golden):
    """float32 torch (NCHW, oneDNN) vs float64 numpy: <= 1e-3 deg, the north-star bar."""
    crops = golden["crops"]
    tw = TorchWHENet(weights)
    y, p, r = tw.get_angle(crops.copy())
    ang = np.stack([y, p, r], axis=1)
    err = np.abs(ang - golden["expected"]["angles"]).max()
    assert err < 1e-3, err
    lg = np.concatenate(tw.predict_logits(O.normalise(crops)), axis=1)
    am = O.argmax_bins(lg)
    safe = golden["expected"]["margins"] > 5e-3
    assert np.array_equal(am[safe], golden["expected"]["argmax"][safe])


def test_numpy_f32_noise_floor(weights, golden):
    r = O.forward(golden["crops"][:4], weights, np.float32)
    ang = np.stack([r["yaw"], r["pitch"], r["roll"]], axis=1)
    assert np.abs(ang - golden["expected"]["angles"][:4]).max() < 1e-3


def test_normalise_is_reference_arithmetic():
    """whenet.py:23-26 literally, on every byte value; LUT == normalise()."""
    v = np.arange(256, dtype=np.uint8)
    img = np.zeros((1, 224, 224, 3), np.uint8)
    img[0, 0, :256 - 32, :] = v[:224, None]
    img[0, 1, :32, :] = v[224:, None]
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]
    ref = ((img / 255 - mean) / std).astype(np.float32)
    got = O.normalise(img)
    assert np.array_equal(ref, got)
    lut = O.normalise_lut()
    assert np.array_equal(lut[np.arange(3)[None, None, None, :], img], got)


def test_decode_is_reference_arithmetic():
    """whenet.py:28-33 + utils.py:7-11 literally, float32 like the reference."""
    rng = np.random.default_rng(3)
    lg = rng.normal(0, 4, size=(5, 252)).astype(np.float32)

    def ref_softmax(x):
        x -= np.max(x, axis=1, keepdims=True)
        a = np.exp(x)
        b = np.sum(np.exp(x), axis=1, keepdims=True)
        return a / b

    idx = np.arange(66, dtype=np.float32)
    idy = np.arange(120, dtype=np.float32)
    yaw = np.sum(ref_softmax(lg[:, :120].copy()) * idy, axis=1) * 3 - 180
    pitch = np.sum(ref_softmax(lg[:, 120:186].copy()) * idx, axis=1) * 3 - 99
    roll = np.sum(ref_softmax(lg[:, 186:].copy()) * idx, axis=1) * 3 - 99
    y, p, r = O.decode(lg)
    for a, b in ((y, yaw), (p, pitch), (r, roll)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)
    assert y.min() >= -180 and y.max() <= 177 and p.min() >= -99 and p.max() <= 96


def test_conv_same_padding_against_torch():
    """TF-SAME asymmetric pads (bottom/right only for even sizes at stride 2)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    for (h, k, s) in ((8, 3, 2), (10, 5, 2), (7, 5, 1), (6, 3, 1), (9, 3, 2)):
        x = rng.normal(size=(2, h, h, 5))
        w = rng.normal(size=(k, k, 5, 1))
        got = O.depthwise(x, w, s)
        out, pb, pa = spec.same_pad(h, k, s)
        xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pb, pa, pb, pa))
        wt = torch.from_numpy(w).permute(2, 3, 0, 1)
        ref = F.conv2d(xt, wt, stride=s, groups=5).permute(0, 2, 3, 1).numpy()
        assert got.shape == (2, out, out, 5)
        np.testing.assert_allclose(got, ref, atol=1e-12)
        # a one-pixel shift of the padding must be detectable
        if pb != pa:
            xt2 = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pa, pb, pa, pb))
            bad = F.conv2d(xt2, wt, stride=s, groups=5).permute(0, 2, 3, 1).numpy()
            assert np.abs(bad - got).max() > 1e-3


def test_hf_efficientnet_cross_check(weights, golden):
    """Independent implementation: transformers' EfficientNet (B0 config) with our weights
    must give the same 7x7x1280 feature map as oracle.backbone()."""
    tr = pytest.importorskip("transformers")
    import torch
    cfg = tr.EfficientNetConfig(
        num_channels=3, image_size=224, width_coefficient=1.0, depth_coefficient=1.0, depth_divisor=8,
        kernel_sizes=[3, 3, 5, 3, 5, 5, 3], in_channels=[32, 16, 24, 40, 80, 112, 192],
        out_channels=[16, 24, 40, 80, 112, 192, 320], depthwise_padding=[],
        strides=[1, 2, 2, 2, 1, 2, 1], num_block_repeats=[1, 2, 2, 3, 3, 4, 1],
        expand_ratios=[1, 6, 6, 6, 6, 6, 6], squeeze_expansion_ratio=0.25, hidden_act="swish",
        hidden_dim=1280, pooling_type="mean", batch_norm_eps=1e-3, batch_norm_momentum=0.99,
        dropout_rate=0.2, drop_connect_rate=0.2)
    model = tr.EfficientNetModel(cfg).eval().double()
    sd = model.state_dict()

    def conv(name):      # HWIO -> OIHW
        return torch.from_numpy(np.transpose(weights[name], (3, 2, 0, 1)).copy()).double()

    def dwk(name):       # (kh,kw,C,1) -> (C,1,kh,kw)
        return torch.from_numpy(np.transpose(weights[name], (2, 3, 0, 1)).copy()).double()

    def vec(name):
        return torch.from_numpy(weights[name].copy()).double()

    new = {}

    def put_bn(dst, src):
        new[f"{dst}.weight"] = vec(f"{src}/gamma")
        new[f"{dst}.bias"] = vec(f"{src}/beta")
        new[f"{dst}.running_mean"] = vec(f"{src}/mean")
        new[f"{dst}.running_var"] = vec(f"{src}/var")

    new["embeddings.convolution.weight"] = conv("stem/conv/kernel")
    put_bn("embeddings.batchnorm", "stem/bn")
    for i, b in enumerate(spec.blocks()):
        p, q = f"encoder.blocks.{i}", f"b{b.index}"
        if b.has_expand:
            new[f"{p}.expansion.expand_conv.weight"] = conv(f"{q}/expand/kernel")
            put_bn(f"{p}.expansion.expand_bn", f"{q}/expand_bn")
        new[f"{p}.depthwise_conv.depthwise_conv.weight"] = dwk(f"{q}/dw/kernel")
        put_bn(f"{p}.depthwise_conv.depthwise_norm", f"{q}/dw_bn")
        new[f"{p}.squeeze_excite.reduce.weight"] = conv(f"{q}/se_reduce/kernel")
        new[f"{p}.squeeze_excite.reduce.bias"] = vec(f"{q}/se_reduce/bias")
        new[f"{p}.squeeze_excite.expand.weight"] = conv(f"{q}/se_expand/kernel")
        new[f"{p}.squeeze_excite.expand.bias"] = vec(f"{q}/se_expand/bias")
        new[f"{p}.projection.project_conv.weight"] = conv(f"{q}/project/kernel")
        put_bn(f"{p}.projection.project_bn", f"{q}/project_bn")
    new["encoder.top_conv.weight"] = conv("head/conv/kernel")
    put_bn("encoder.top_bn", "head/bn")
    missing = [k for k in sd if k not in new and "num_batches_tracked" not in k]
    assert not missing, missing[:5]
    for k, v in new.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    model.load_state_dict(new, strict=False)

    crops = golden["crops"][:2]
    x = O.normalise(crops).astype(np.float64)
    ours = O.backbone(x, weights)
    with torch.no_grad():
        hf = model(torch.from_numpy(x).permute(0, 3, 1, 2)).last_hidden_state.permute(0, 2, 3, 1).numpy()
    assert hf.shape == ours.shape == (2, 7, 7, 1280)
    np.testing.assert_allclose(hf, ours, rtol=0, atol=1e-9)


def test_oracle_class_surface(weights, golden):
    m = O.OracleWHENet(weights)
    y, p, r = m.get_angle(golden["crops"][:1])
    assert y.shape == p.shape == r.shape == (1,)
    with pytest.raises(ValueError):
        m.get_angle(np.zeros((224, 224, 3), np.uint8))
