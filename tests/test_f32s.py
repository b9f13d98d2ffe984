"""WHENET_F32S (round 5): float32 storage and accumulation, the 1x1 products as binary16 hi/lo pairs on the f16 matrix
cores (pw.hip PwOps<float, true>).  Held to the SAME bar as the exact-f32 configuration: <= 1e-3 deg and equal argmax
against the float64 oracle on the 512-crop set and against the reference-run fixture; per-kernel tolerance 2e-5; batch
invariance bitwise."""
import os

import numpy as np
import pytest

from oracle import whenet_oracle as O
from tests import refcases as C
from whenet_hip import _lib, spec, synth, weights as W

pytestmark = pytest.mark.gpu
SE_FUSE_DEFAULT = 1      # option se_fuse: the project GEMM computes the gate where that pays (blocks 4-6)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def blob():
    return W.pack(W.synthetic(1234))


@pytest.fixture(scope="module")
def hs(blob):
    h = _lib.Handle(blob, device=0, dtype=_lib.F32S)
    yield h
    h.close()


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_info_reports_float_storage(hs):
    assert hs.info().dtype == _lib.F32
    with pytest.raises(ValueError):
        with _lib.Handle(W.pack(W.synthetic(1234)), device=0, dtype=_lib.F16) as h16:
            h16.set_option("split_pw", 1)


def test_f32s_meets_the_parity_bar_on_512_crops(hs, blob):
    fx = np.load(os.path.join(GOLD, "f16_set512_expected.npz"))
    crops = np.concatenate([synth.scene_crops(256, seed=41), synth.noise_crops(256, seed=42)])
    y, a, l = hs.forward(crops)
    e = np.abs(y - fx["angles"])
    safe = fx["margins"] > 2e-3
    with _lib.Handle(blob, device=0, dtype=_lib.F32) as h32:
        y32, a32, l32 = h32.forward(crops)
    e32 = np.abs(y32 - fx["angles"])
    print(f"\n[f32s, 512 crops vs oracle] max {e.max():.2e} mean {e.mean():.2e} deg, max |logit err| {np.abs(l - fx['logits']).max():.2e}"
          f"   (exact f32: max {e32.max():.2e} mean {e32.mean():.2e} deg, logits {np.abs(l32 - fx['logits']).max():.2e})")
    assert e.max() <= 1e-3
    assert np.array_equal(a[safe], fx["argmax"][safe])
    # split_pw = 0 on the same handle: the exact-f32 kernels, bitwise the WHENET_F32 handle
    hs.set_option("split_pw", 0)
    try:
        y0, a0, l0 = hs.forward(crops[:70])
    finally:
        hs.set_option("split_pw", 1)
    assert np.array_equal(y0, y32[:70]) and np.array_equal(l0, l32[:70])


def test_f32s_against_the_reference_run(hs):
    ref = dict(np.load(os.path.join(GOLD, "reference_get_angle.npz")))
    crops = C.crops64()
    y, a, l = hs.forward(crops)
    assert np.abs(y - ref["n64_angles"]).max() <= 1e-3
    safe = O.top2_margin(ref["n64_logits"]) > 2e-3
    assert np.array_equal(a[safe], O.argmax_bins(ref["n64_logits"])[safe])


@pytest.mark.parametrize("index", [1, 2, 3, 6, 9, 12, 13, 16])
def test_f32s_block_kernels_within_f32_tolerance(hs, blob, index):
    """every pointwise kernel form (tile / split-K x gate from memory / fused gate x skip) on the oracle's own block input"""
    taps = {}
    crops = np.load(os.path.join(GOLD, "golden_crops.npy"))[:3]
    O.forward(crops, W.synthetic(1234), np.float64, taps=taps)
    b = spec.blocks()[index - 1]
    x = (taps["stem"] if index == 1 else taps[f"b{index - 1}/out"]).astype(np.float32)
    for se_fuse in (0, 2):
        hs.set_option("se_fuse", se_fuse)
        try:
            r = hs.op_block(index, x)
        finally:
            hs.set_option("se_fuse", SE_FUSE_DEFAULT)
        assert rel_err(r["out"], taps[f"b{index}/out"]) < 6e-5, (index, se_fuse)
    hs.set_option("fuse_front", 0)           # the expand conv as a pointwise launch of its own (tile kernel, K = 16..192)
    try:
        r = hs.op_block(index, x)
    finally:
        hs.set_option("fuse_front", 1)
    if b.has_expand:
        assert rel_err(r["expand"], taps[f"b{index}/expand"]) < 2e-5
    assert rel_err(r["out"], taps[f"b{index}/out"]) < 6e-5


@pytest.mark.parametrize("index", list(range(2, 13)))
def test_front2s_kernel_on_every_block_shape(hs, index):
    """Round 6: front2s.hip -- expand as binary16 hi/lo products with pixels as MFMA rows, depthwise taps as per-channel Toeplitz
    products on the matrix cores (exact float32 on v_mfma_f32_4x4x1 or hi/lo pairs on v_mfma_f32_4x4x4_f16, per layer), the
    expanded tile at float32 precision in LDS.  Option front_impl=2 runs it on every block it exists for (2-12); held to the
    per-kernel float32 tolerance against the oracle's depthwise output and to front.hip's own result."""
    taps = {}
    crops = np.load(os.path.join(GOLD, "golden_crops.npy"))[:3]
    O.forward(crops, W.synthetic(1234), np.float64, taps=taps)
    x = taps[f"b{index - 1}/out"].astype(np.float32)
    outs = {}
    for impl in (0, 2):
        hs.set_option("front_impl", impl)
        hs.set_option("se_fuse", 0)
        try:
            outs[impl] = hs.op_block(index, x)
        finally:
            hs.set_option("front_impl", 1)
            hs.set_option("se_fuse", SE_FUSE_DEFAULT)
    r = outs[2]
    assert not np.array_equal(r["dw"], outs[0]["dw"]), "front_impl=2 did not change the kernel"
    assert rel_err(r["dw"], taps[f"b{index}/dw"]) < 2e-5, "dw"
    assert rel_err(r["dw"], outs[0]["dw"]) < 2e-5, "front2s vs front.hip"
    assert rel_err(r["gate"], taps[f"b{index}/gate"].reshape(r["gate"].shape)) < 4e-5, "gate"
    assert rel_err(r["out"], taps[f"b{index}/out"]) < 6e-5, "out"


def test_front2s_everywhere_meets_the_bar_and_is_batch_invariant(hs):
    fx = np.load(os.path.join(GOLD, "f16_set512_expected.npz"))
    crops = np.concatenate([synth.scene_crops(256, seed=41), synth.noise_crops(256, seed=42)])[:128]
    ydef, _, ldef = hs.forward(crops)
    hs.set_option("front_impl", 2)
    try:
        y, a, l = hs.forward(crops)
        for n in (1, 3, 17, 64):
            yn, an, ln = hs.forward(crops[:n])
            assert np.array_equal(ln, l[:n]), n
    finally:
        hs.set_option("front_impl", 1)
    assert np.abs(y - fx["angles"][:128]).max() <= 1e-3
    safe = fx["margins"][:128] > 2e-3
    assert np.array_equal(a[safe], fx["argmax"][:128][safe])
    assert not np.array_equal(l, ldef)
    hs.set_option("front_impl", 0)               # round 5's schedule: front.hip on every block
    try:
        y0, a0, l0 = hs.forward(crops)
    finally:
        hs.set_option("front_impl", 1)
    assert np.abs(y0 - fx["angles"][:128]).max() <= 1e-3 and not np.array_equal(l0, ldef)


def test_f32s_fold12(hs):
    """Round 6: block 1's project conv folded into block 2's expand weights for f32s too (front.hip's split form multiplies its
    float32 operand by block 1's gate before the hi/lo split): one launch less, the same bar, another rounding path."""
    fx = np.load(os.path.join(GOLD, "f16_set512_expected.npz"))
    crops = np.concatenate([synth.scene_crops(256, seed=41), synth.noise_crops(256, seed=42)])[:96]
    k1 = hs.info().n_kernels_per_forward
    y1, a1, l1 = hs.forward(crops)
    hs.set_option("fold12", 0)
    try:
        k0 = hs.info().n_kernels_per_forward
        y0, a0, l0 = hs.forward(crops)
    finally:
        hs.set_option("fold12", 1)
    assert k1 == k0 - 1
    assert np.abs(y1 - fx["angles"][:96]).max() <= 1e-3 and np.abs(y0 - fx["angles"][:96]).max() <= 1e-3
    assert not np.array_equal(l0, l1) and np.abs(l0 - l1).max() < 2e-4
    assert np.array_equal(hs.forward(crops[7:8])[2][0], l1[7])


def test_f32s_head_conv_and_pooling(hs):
    taps = {}
    crops = np.load(os.path.join(GOLD, "golden_crops.npy"))[:5]
    w = W.synthetic(1234)
    r = O.forward(crops, w, np.float64, taps=taps)
    x = taps["b16/out"].astype(np.float32)
    got = hs.op_head(x)
    feat = O.swish(O.batchnorm(O.conv2d(taps["b16/out"], w["head/conv/kernel"], 1), w, "head/bn")).mean(axis=(1, 2)) if hasattr(O, "batchnorm") else None
    if feat is not None:
        assert rel_err(got["feat"], feat) < 2e-5
    assert np.abs(got["logits"] - r["logits"][:5]).max() < 5e-4
    for head7 in (1, 0):                                   # the fused head conv + pooling kernel and the split-K GEMM + heads kernel
        hs.set_option("head_fuse", head7)
        try:
            g2 = hs.op_head(x)
        finally:
            hs.set_option("head_fuse", 1)
        assert np.abs(g2["logits"] - r["logits"][:5]).max() < 5e-4


def test_f32s_batch_invariance_bitwise(hs):
    crops = np.concatenate([synth.scene_crops(40, seed=3), synth.noise_crops(37, seed=4)])
    y, a, l = hs.forward(crops)
    for n in (1, 2, 3, 16, 17, 21, 64):
        yn, an, ln = hs.forward(crops[:n])
        assert np.array_equal(yn, y[:n]) and np.array_equal(ln, l[:n]), n
    hs.set_option("lanes", 1)
    try:
        y1, _, l1 = hs.forward(crops)
    finally:
        hs.set_option("lanes", 2)
    assert np.array_equal(y1, y) and np.array_equal(l1, l)


def test_f32s_dropin_class():
    import whenet
    ref = dict(np.load(os.path.join(GOLD, "reference_get_angle.npz")))
    with whenet.WHENet(dtype="f32s") as m:
        y, p, r = m.get_angle(C.crops64()[:9])
    assert np.abs(np.stack([y, p, r], 1) - ref["n9_angles"]).max() <= 1e-3


def test_f32s_is_the_class_default_and_leaves_binary16_range_loudly():
    """Round 6: WHENet(snapshot) with no dtype is the f32s configuration (the reference's callers pass none: demo.py:20,
    demo_video.py:40).  Its precondition is activations inside binary16's range; a snapshot that breaks it (here: the stem's BatchNorm
    scale times 1e6) gives NaN angles on the handle -- never a silently wrong number -- and the class then runs the exact-float32
    kernels on the same handle: bitwise a dtype='f32' model, with a warning."""
    import whenet
    ref = dict(np.load(os.path.join(GOLD, "reference_get_angle.npz")))
    crops = C.crops64()[:9]
    with whenet.WHENet() as m, whenet.WHENet(dtype="f32s") as ms:
        assert m._dtype == _lib.F32S
        got = m.get_angle(crops)
        assert np.abs(np.stack(got, 1) - ref["n9_angles"]).max() <= 1e-3
        assert all(np.array_equal(a, b) for a, b in zip(got, ms.get_angle(crops)))
    w = W.synthetic(1234)
    w = dict(w)
    w["stem/bn/gamma"] = w["stem/bn/gamma"] * np.float32(1e6)
    big = W.pack(w)
    with _lib.Handle(big, device=0, dtype=_lib.F32S) as h:
        assert np.isnan(h.forward(crops[:2])[0]).any()                 # the handle alone: NaN, not a wrong number
    with whenet.WHENet(big, dtype="f32") as m32:
        want = m32.get_angle(crops[:4])
    assert all(np.isfinite(a).all() for a in want)
    with whenet.WHENet(big) as m:
        with pytest.warns(RuntimeWarning, match="binary16"):
            got = m.get_angle(crops[:4])
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert m._dtype == _lib.F32
        got2 = m.get_angle(crops[:4])                                   # (later calls: no warning, the same bits)
        assert all(np.array_equal(a, b) for a, b in zip(got2, want))


@pytest.mark.parametrize("dtype", ["f16", "f32s"])
def test_staged_splitk_option(blob, dtype):
    """Option pw_staged (round 5): the K >= 1152 / 14x14 K = 672 project GEMMs fetch their activation rows coalesced through per-wave
    LDS.  Another summation order than the direct kernel (wave p sums 128-byte k-groups instead of interleaved k-steps): results
    agree to rounding, each form is bitwise batch-invariant on its own, and the choice never depends on the batch."""
    dt = {"f16": _lib.F16, "f32s": _lib.F32S}[dtype]
    crops = np.concatenate([synth.scene_crops(30, seed=8), synth.noise_crops(27, seed=9)])
    outs = {}
    with _lib.Handle(blob, device=0, dtype=dt) as h:
        for staged in (1, 0):
            h.set_option("pw_staged", staged)
            y, a, l = h.forward(crops)
            for n in (1, 3, 20, 21):                      # both tile shapes of the split-K kernels (B2 = 1 below 21 crops on 14x14)
                yn, an, ln = h.forward(crops[:n])
                assert np.array_equal(yn, y[:n]) and np.array_equal(ln, l[:n]), (staged, n)
            outs[staged] = (y, l)
    tol = 0.2 if dtype == "f16" else 2e-4
    assert np.abs(outs[0][0] - outs[1][0]).max() < tol
    assert not np.array_equal(outs[0][1], outs[1][1]), "pw_staged did not change the kernel"
