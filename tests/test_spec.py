"""Structural invariants of the restated graph (SURVEY.md §8c "how parity will be pinned
instead", item 2): parameter counts, array count, MAC total, TF-SAME padding."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import b0_spec as G
from whenet_hip import spec


def test_param_counts():
    assert spec.param_count() == (4_049_564, 322_812)
    assert len(spec.tensors()) == 315
    assert len(spec.bn_names()) == 49


def test_weighted_layer_census():
    names = [t.name for t in spec.tensors()]
    convs = [n for n in names if n.endswith("/kernel") and not n.split("/")[0] in ("yaw", "pitch", "roll")]
    assert len(convs) == 81                       # 33 Conv2D + 16 depthwise + 32 SE convs
    assert sum(n.endswith("/dw/kernel") for n in names) == 16
    assert sum("/se_" in n and n.endswith("/bias") for n in names) == 32


def test_mac_total_and_classes():
    w = spec.work_table()
    assert sum(v.macs for v in w.values()) == spec.TOTAL_MACS == 384_857_312
    assert w["pw"].macs == 338_537_472
    assert w["dw"].macs == 34_532_064
    assert w["stem"].macs == 10_838_016
    assert w["se"].macs == 627_200
    assert w["fc"].macs == 322_560
    assert (w["dw"].in_elems, w["dw"].out_elems) == (3_788_288, 2_306_528)
    assert (w["pw"].in_elems, w["pw"].out_elems) == (2_886_688, 4_029_760)


def test_block_table():
    b = spec.blocks()
    assert len(b) == 16
    assert [x.h_out for x in b] == [112, 56, 56, 28, 28, 14, 14, 14, 14, 14, 14, 7, 7, 7, 7, 7]
    assert [x.index for x in b if x.has_skip] == [3, 5, 7, 8, 10, 11, 13, 14, 15]
    assert [x.se_reduced for x in b] == [8, 4, 6, 6, 10, 10, 20, 20, 20, 28, 28, 28, 48, 48, 48, 48]


def test_same_pad_known_cases():
    # even input, stride 2: k3 -> (0,1), k5 -> (1,2); stride 1: k3 -> (1,1), k5 -> (2,2)
    assert spec.same_pad(224, 3, 2) == (112, 0, 1)
    assert spec.same_pad(112, 3, 2) == (56, 0, 1)
    assert spec.same_pad(56, 5, 2) == (28, 1, 2)
    assert spec.same_pad(14, 5, 2) == (7, 1, 2)
    assert spec.same_pad(28, 3, 2) == (14, 0, 1)
    assert spec.same_pad(56, 3, 1) == (56, 1, 1)
    assert spec.same_pad(7, 5, 1) == (7, 2, 2)


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 300), st.sampled_from([1, 3, 5, 7]), st.sampled_from([1, 2, 3]))
def test_same_pad_properties(n, k, s):
    out, pb, pa = spec.same_pad(n, k, s)
    assert out == -(-n // s)
    assert 0 <= pb <= pa <= pb + 1
    # the last window fits exactly inside the padded input, and no padding is wasted
    need = (out - 1) * s + k
    assert pb + pa == max(need - n, 0)


def test_oracle_geometry_is_independent_and_agrees():
    """oracle/b0_spec.py decodes efficientnet's own block strings; whenet_hip/spec.py carries the
    product's table (and csrc/spec.h a third copy, compared through the C ABI in
    tests/test_capi_cpu.py).  The checker must not import the product's geometry -- and the tables
    must agree."""
    import os
    import re
    here = os.path.dirname(os.path.abspath(G.__file__))
    for f in os.listdir(here):
        if f.endswith(".py"):
            src = open(os.path.join(here, f)).read()
            assert not re.search(r"^\s*(from|import)\s+(whenet_hip|whenet)\b", src, re.M), f
    ours, theirs = G.mbconv_blocks(), spec.blocks()
    assert len(ours) == len(theirs) == 16
    for a, b in zip(ours, theirs):
        assert (a.number, a.kernel, a.stride, a.expand, a.filters_in, a.filters_out, a.size_in, a.size_out) == \
               (b.index, b.k, b.s, b.expand, b.cin, b.cout, b.h_in, b.h_out)
        assert (a.filters_mid, a.se_width, a.expands, a.identity_skip) == (b.cexp, b.se_reduced, b.has_expand, b.has_skip)
        assert G.tf_same(a.size_in, a.kernel, a.stride) == spec.same_pad(b.h_in, b.k, b.s)
    assert G.BN_EPSILON == spec.BN_EPS and G.INPUT_SIZE == spec.IMG
    assert tuple(G.IMAGENET_MEAN) == spec.MEAN and tuple(G.IMAGENET_STD) == spec.STD
    assert tuple(n for _, n in G.BINS) == (spec.N_YAW, spec.N_PITCH, spec.N_ROLL)


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 300), st.sampled_from([1, 3, 5, 7]), st.sampled_from([1, 2, 3]))
def test_two_same_pad_statements_agree(n, k, s):
    assert G.tf_same(n, k, s) == spec.same_pad(n, k, s)
