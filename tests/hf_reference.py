"""HuggingFace ``transformers`` EfficientNet (an independent third-party implementation of B0) loaded
with WHENet-layout weights.  TEST INFRASTRUCTURE: used by tests/test_oracle.py (live cross-check on
the build box) and by tests/golden/make_hf_fixture.py (the committed fixture the GPU box checks
against, where running transformers is not required)."""
import numpy as np


def hf_backbone_features(weights, x_nhwc_f64):
    """[N,224,224,3] normalised float64 -> [N,7,7,1280] float64 from transformers' EfficientNetModel."""
    import torch
    import transformers as tr
    from oracle import b0_spec as G
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
    for i, b in enumerate(G.mbconv_blocks()):
        p, q = f"encoder.blocks.{i}", f"b{b.number}"
        if b.expands:
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
    with torch.no_grad():
        return model(torch.from_numpy(x_nhwc_f64).permute(0, 3, 1, 2)).last_hidden_state.permute(0, 2, 3, 1).numpy()
