"""Parity tests proper: the HIP path, called through the C ABI, against the float64 oracle on
the same seeded inputs (pytest -m gpu, on the MI355X box).

Tolerances (stated here, per dtype):
  f32  angles <= 1e-3 deg (the north-star bar), bin argmax equal wherever the oracle's top-2
       logit margin exceeds 2e-3 (float32 round-off moves logits by ~3e-4, see
       tests/golden/golden.json noise_floor_*), per-kernel tensors <= 2e-5 * scale;
  f16  cannot meet 1e-3 deg (10-bit mantissa through 82 conv layers).  Its error is rounding NOISE with a long tail:
       over 1536 angles (512 crops) mean 0.048, p95 0.21, p99 0.47, max 1.2 deg, and WHICH angle is the outlier
       changes with every change of a rounding point (round 2's own error study, profiles/r02/f16_error_study.txt,
       has maxima from 0.42 to 0.90 deg on the same 48 crops across rounding subsets).  So the contract is stated on
       the distribution AGAINST THE FLOAT64 ORACLE (test_f16_accuracy_contract: 48 crops; round 4:
       test_f16_error_distribution_512_crops: 512 crops against tests/golden/f16_set512_expected.npz, not against the
       library's own f32 path): mean / p95 / p99 / p99.9 / bin flips; every single f16 angle of the small sets (golden
       crops, 48-crop set, batch tests) is within F16_DEG = 1.0 deg, and the 512-crop set's maximum -- the tail of 1536
       draws -- within F16_DEG_TAIL = 1.5 deg = half a 3-degree bin;
       per-kernel tensors <= 1.5e-2 * scale (each kernel alone, fed oracle inputs rounded to f16);
       bin argmax: asserted on EVERY golden crop -- equal, or the oracle's top-2
       margin is below the measured logit error (the synthetic heads are 3-bin-wide Gaussian bumps:
       neighbouring bins differ by <= 0.38, so f16 may legitimately pick the neighbour) and the
       picked bin is the oracle's runner-up; at most F16_ARGMAX_FLIPS of the 24 may flip.
"""
import os

import numpy as np
import pytest

from oracle import whenet_oracle as O
from whenet_hip import _lib, spec, synth, weights as W

pytestmark = pytest.mark.gpu
SE_FUSE_DEFAULT = 1      # option se_fuse: the project GEMM computes the gate where that pays (blocks 4-6)

F32_DEG = 1e-3
F16_DEG = 1.0              # every f16 angle of the small sets; round 3 had widened this to 1.5 for all sets
F16_DEG_TAIL = 1.5         # the maximum over the 512-crop set only (1536 draws of the noise's tail; p99.9 is bounded below it)
F16_ARGMAX_FLIPS = 3
MARGIN_F32 = 2e-3
DTYPES = [("f32", _lib.F32), ("f16", _lib.F16)]


@pytest.fixture(scope="module")
def blob(weights):
    return W.pack(weights)


@pytest.fixture(scope="module", params=DTYPES, ids=[d[0] for d in DTYPES])
def handle(request, blob):
    h = _lib.Handle(blob, device=0, dtype=request.param[1])
    h.name = request.param[0]
    yield h
    h.close()


@pytest.fixture(scope="module")
def taps(weights, golden):
    """Oracle intermediates (float64) for 2 crops: one sample crop, one scene crop."""
    crops = golden["crops"][[0, 3]]
    t = {}
    x = O.normalise(crops).astype(np.float64)
    f = O.backbone(x, weights, taps=t)
    t["crops"] = crops
    t["logits"] = O.heads(f, weights)
    return t


def rel_err(got, ref):
    """max over elements of |got - ref| / (|ref| + rms(ref)): relative for large values,
    absolute (in units of the tensor's rms) for small ones."""
    ref = np.asarray(ref, np.float64)
    rms = max(np.sqrt((ref ** 2).mean()), 1e-6)
    return float((np.abs(np.asarray(got, np.float64) - ref) / (np.abs(ref) + rms)).max())


def tol(h):
    # f32: a few 1e-6 measured (tools/gpu_diag.py); f16: 2^-11 input/weight/output roundings
    # through a k*k*C / K-deep sum, a few 1e-3 .. 1e-2 measured
    return 2e-5 if h.name == "f32" else 1.5e-2


def test_info(handle):
    i = handle.info()
    assert i.abi_version == _lib.ABI_VERSION
    assert (i.params_backbone, i.params_heads, i.n_tensors) == (4_049_564, 322_812, 315)
    # stem, dw(b1), 15 front, 16 se, 16 project, head conv, heads = 51 with option se_fuse=0; by default the project
    # GEMMs of the blocks where it pays compute their squeeze-excite gate themselves (f16: blocks 4-6; f32: by the same
    # rule on front.hip's tile plans); 36 with se_fuse=2 (every block with a fused front kernel)
    # f16 handles drop block 1's project launch (option fold12: folded into block 2's expand weights)
    assert i.macs_per_crop == spec.TOTAL_MACS and 35 <= i.n_kernels_per_forward <= 50
    # ... and the stem launch (option stem_fuse: computed inside block 1's depthwise kernel, stemdw.hip)
    folded = 1 if handle.name == "f16" else 0
    stemdw = 1
    # round 6, option mb7 (off by default): blocks 13-16 of an f16 handle as one launch each instead of front + squeeze-excite + project
    mb = 4 if handle.name == "f16" else 0
    try:
        handle.set_option("se_fuse", 0)
        assert handle.info().n_kernels_per_forward == 51 - folded - stemdw
        handle.set_option("mb7", 1)
        assert handle.info().n_kernels_per_forward == 51 - folded - stemdw - 2 * mb
        handle.set_option("se_fuse", 2)
        assert handle.info().n_kernels_per_forward == 36 - folded - stemdw - mb
        handle.set_option("mb7", 0)
        assert handle.info().n_kernels_per_forward == 36 - folded - stemdw
        handle.set_option("fold12", 0)
        assert handle.info().n_kernels_per_forward == 36 - stemdw
    finally:
        handle.set_option("mb7", 0)
        handle.set_option("fold12", 1)
        handle.set_option("se_fuse", SE_FUSE_DEFAULT)
    assert b"gfx950" in i.arch and i.compute_units >= 200


def test_stem_kernel(handle, taps):
    got = handle.op_stem(taps["crops"])
    assert rel_err(got, taps["stem"]) < tol(handle)


def test_f32_stem_on_the_matrix_cores_has_the_scalar_kernels_bits(tmp_path, golden):
    """Round 4: the f32 stem runs on v_mfma_f32_32x32x2_f32 (exact f32: an fmaf chain in k order).  Its output must be
    BITWISE the scalar kernel's (WHENET_STEM_SCALAR_F32=1, read once per process: two child processes)."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = ("import sys, numpy as np; sys.path.insert(0, sys.argv[1] + '/headposeestimation-whenet_amd'); sys.path.insert(0, sys.argv[1]);"
            "import torch; from whenet_hip import _lib, weights as W;"
            "h = _lib.Handle(W.pack(W.synthetic(1234)), device=0, dtype=_lib.F32);"
            "c = np.load(sys.argv[1] + '/tests/golden/golden_crops.npy')[:3];"
            "np.savez(sys.argv[2], stem=h.op_stem(c), logits=h.forward(c)[2]); h.close()")
    outs = []
    for flag in ("0", "1"):
        f = str(tmp_path / f"stem{flag}.npz")
        env = dict(os.environ, WHENET_STEM_SCALAR_F32=flag)
        r = subprocess.run([sys.executable, "-c", code, root, f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(dict(np.load(f)))
    assert np.array_equal(outs[0]["stem"], outs[1]["stem"]) and np.array_equal(outs[0]["logits"], outs[1]["logits"])


@pytest.mark.parametrize("index", list(range(1, 17)))
def test_mbconv_block_kernels(handle, taps, index):
    """expand GEMM, depthwise, SE gate, project GEMM (+skip) of every block, each on the
    oracle's own input for that block: covers every distinct layer shape of the network.  The fused
    expand+depthwise stage is checked in all its forms: front.hip (option front_impl=0: bitwise the two-launch
    schedule), and for f16 front2.hip (front_impl=2: depthwise taps as Toeplitz products on the matrix cores, f16
    tap weights -- another summation order, so within the kernel tolerance of the oracle) and the default per-layer
    choice between the two."""
    b = spec.blocks()[index - 1]
    x = taps["stem"] if index == 1 else taps[f"b{index - 1}/out"]
    t = tol(handle)
    p = f"b{index}"
    if b.has_expand:
        # the two-launch schedule materialises the expanded tensor: check it, then front.hip
        # must reproduce the same depthwise output bitwise
        handle.set_option("fuse_front", 0)
        try:
            r0 = handle.op_block(index, x.astype(np.float32))
        finally:
            handle.set_option("fuse_front", 1)
        assert rel_err(r0["expand"], taps[f"{p}/expand"]) < t, "expand"
        assert rel_err(r0["dw"], taps[f"{p}/dw"]) < 2 * t, "dw (unfused)"
        assert rel_err(r0["out"], taps[f"{p}/out"]) < 3 * t, "out (unfused)"
    impls = (0, 2, 1) if (b.has_expand and handle.name == "f16") else (1,)
    for impl in impls:
        handle.set_option("front_impl", impl)
        handle.set_option("se_fuse", 0)                    # (a squeeze-excite launch writes the gate: it can be checked)
        try:
            r = handle.op_block(index, x.astype(np.float32))
            handle.set_option("se_fuse", 2)                # the project GEMM computes the gate itself (every block)
            rf = handle.op_block(index, x.astype(np.float32))
        finally:
            handle.set_option("front_impl", 1)
            handle.set_option("se_fuse", SE_FUSE_DEFAULT)
        # the fused prologue's arithmetic is the stand-alone kernel's: the same block output, bit for bit
        assert np.array_equal(rf["out"], r["out"]) and np.array_equal(rf["dw"], r["dw"]), f"se_fuse changes bits (front_impl={impl})"
        assert not b.has_expand or np.isnan(rf["gate"]).all()          # (no launch wrote a gate)
        if b.has_expand:
            if impl == 0 or handle.name == "f32":
                assert np.array_equal(r["dw"], r0["dw"]), "fused expand+depthwise differs from pw+dw"
            else:
                # the same f16 expanded tensor, f16 instead of f32 tap weights: a rounding apart
                assert rel_err(r["dw"], r0["dw"]) < t, f"front2 vs pw+dw (front_impl={impl})"
            # the fused kernel also applies the SE reduce conv to its channel sums (another summation
            # order than se.hip's): same gate and block output within the kernel tolerance
            assert rel_err(r["gate"], r0["gate"]) < 2 * t and rel_err(r["out"], r0["out"]) < 3 * t
        # the stages below consume the kernel's own upstream output, so errors chain a little
        assert rel_err(r["dw"], taps[f"{p}/dw"]) < 2 * t, f"dw (front_impl={impl})"
        assert rel_err(r["gate"], taps[f"{p}/gate"].reshape(r["gate"].shape)) < 2 * t, "gate"
        assert rel_err(r["out"], taps[f"{p}/out"]) < 3 * t, "out"


@pytest.mark.parametrize("dt", DTYPES, ids=[d[0] for d in DTYPES])
def test_front7_group_kernel(blob, taps, golden, dt):
    """Round 4: blocks 13-16 (7 x 7 maps) run front7.hip -- a group of crops per workgroup (2 crops up to 16 crops per
    launch, 4 above), the chunk's expand weights staged once in LDS, the image-only tile.  Each block on the oracle's own
    input: within the kernel tolerance of the oracle and of round 3's kernel (option front7=0) -- for f32 the depthwise
    output is BITWISE round 3's (same fmaf chains in the same order; only the grouping of the squeeze-excite sums
    differs); and the group size changes no bit: crops travel through launches of 1, 2, 3, 16, 17 and 21 crops (tail
    groups of both sizes) with identical results."""
    name, dtype = dt
    t = 1.5e-2 if name == "f16" else 2e-5
    with _lib.Handle(blob, device=0, dtype=dtype) as h:
        for index in (13, 14, 15, 16):
            x = taps[f"b{index - 1}/out"].astype(np.float32)
            h.set_option("se_fuse", 0)
            r7 = h.op_block(index, x)
            h.set_option("front7", 0)
            r3 = h.op_block(index, x)
            h.set_option("front7", 1)
            h.set_option("se_fuse", SE_FUSE_DEFAULT)
            if name == "f32":
                assert np.array_equal(r7["dw"], r3["dw"]), "f32 front7: depthwise output differs from front.hip's"
                assert not np.array_equal(r7["gate"], r3["gate"]), "front7 is not active"
            else:
                assert not np.array_equal(r7["dw"], r3["dw"]), "front7 is not active"
            assert rel_err(r7["dw"], r3["dw"]) < t and rel_err(r7["gate"], r3["gate"]) < 2 * t and rel_err(r7["out"], r3["out"]) < 3 * t
            assert rel_err(r7["dw"], taps[f"b{index}/dw"]) < 2 * t and rel_err(r7["out"], taps[f"b{index}/out"]) < 3 * t
            # the same block on a ragged batch of 21: every crop bitwise what it is alone
            xs = np.concatenate([x] * 11)[:21]
            big = h.op_block(index, xs)
            for i in range(21):
                assert np.array_equal(big["out"][i], r7_like(h, index, xs[i:i + 1])), (index, i)
        crops = np.concatenate([golden["crops"], synth.scene_crops(13, seed=77)])          # 21 crops
        y21, a21, l21 = h.forward(crops)
        for lo, hi in ((0, 1), (1, 3), (3, 6), (0, 16), (4, 21), (20, 21)):
            y, a, l = h.forward(crops[lo:hi])
            assert np.array_equal(l, l21[lo:hi]) and np.array_equal(y, y21[lo:hi]), (lo, hi)
        h.set_option("front7", 0)
        y0, a0, l0 = h.forward(crops)
        assert not np.array_equal(l0, l21) and np.abs(l0 - l21).max() < (0.5 if name == "f16" else 2e-3)


def r7_like(h, index, x1):
    return h.op_block(index, x1)["out"][0]


def test_block1_project_folded_into_block2_expand(handle, taps):
    """Option fold12 (f16): block 2's front kernel reads block 1's gated depthwise output through the composed
    project1 x expand2 weights (snapshot.cpp).  The two-step form of the same range is bitwise the per-block operators
    chained; the folded form is the same function with other rounding points: within the kernel tolerance of the
    two-step form and of the oracle's block-2 output.  f32 handles never fold."""
    x = taps["stem"].astype(np.float32)
    t = tol(handle)
    handle.set_option("fold12", 0)
    try:
        two = handle.op_block_range(1, 2, x)
    finally:
        handle.set_option("fold12", 1)
    chained = handle.op_block(2, handle.op_block(1, x)["out"])["out"]
    assert np.array_equal(two, chained)
    one = handle.op_block_range(1, 2, x)
    if handle.name == "f32":
        assert np.array_equal(one, two)
    else:
        assert not np.array_equal(one, two), "fold12 is not active on the f16 handle"
        assert rel_err(one, two) < 2 * t
    assert rel_err(two, taps["b2/out"]) < 4 * t and rel_err(one, taps["b2/out"]) < 4 * t
    # a longer range across the fold, against the oracle's tap
    assert rel_err(handle.op_block_range(1, 4, x), taps["b4/out"]) < 6 * t
    # and ranges that do not hold both blocks are the plain blocks
    assert np.array_equal(handle.op_block_range(2, 2, taps["b1/out"].astype(np.float32)),
                          handle.op_block(2, taps["b1/out"].astype(np.float32))["out"])
    with pytest.raises(ValueError):
        handle.op_block_range(3, 2, taps["b2/out"].astype(np.float32))


def test_head_kernels(handle, taps):
    r = handle.op_head(taps["b16/out"].astype(np.float32))
    t = tol(handle)
    assert rel_err(r["feat"], taps["head"].mean(axis=(1, 2))) < 3 * t
    assert np.abs(r["logits"] - taps["logits"]).max() < (2e-3 if handle.name == "f32" else 1.0)


def test_head_conv_fused_with_pooling(blob, taps, golden):
    """Round 4 ("K-pw fused with GAP"; whenet.py:8-10): the f16 head conv pools its own output (head7.hip) --
    a group of crops per workgroup, every crop on its own two MFMA strips, pooled in f32 before any rounding.  Against the
    oracle's pooled features, against round 3's two stages (option head_fuse=0), and bitwise the same for a crop
    whatever launch it travels in (groups of 2 crops up to 16 per launch, of 4 above: 1, 2, 3, 16, 17, 21 crops)."""
    x = taps["b16/out"].astype(np.float32)
    want = taps["head"].mean(axis=(1, 2))
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        r1 = h.op_head(x)
        h.set_option("head_fuse", 0)
        r0 = h.op_head(x)
        h.set_option("head_fuse", 1)
        assert not np.array_equal(r1["feat"], r0["feat"]), "head_fuse is not active"
        assert rel_err(r1["feat"], want) < 1.5e-2 and rel_err(r0["feat"], want) < 3 * 1.5e-2
        # pooled in f32 before the rounding to f16: closer to the oracle than the two-stage form
        assert np.abs(r1["feat"] - want).mean() <= np.abs(r0["feat"] - want).mean()
        assert np.abs(r1["logits"] - taps["logits"]).max() < 1.0
        xs = np.concatenate([x] * 11)[:21]
        big = h.op_head(xs)
        for i in (0, 1, 15, 16, 17, 20):
            one = h.op_head(xs[i:i + 1])
            assert np.array_equal(big["feat"][i], one["feat"][0]) and np.array_equal(big["logits"][i], one["logits"][0]), i
        crops = np.concatenate([golden["crops"], synth.scene_crops(13, seed=78)])          # 21 crops
        y21, a21, l21 = h.forward(crops)
        for lo, hi in ((0, 1), (1, 3), (0, 16), (4, 21), (20, 21)):
            y, a, l = h.forward(crops[lo:hi])
            assert np.array_equal(l, l21[lo:hi]) and np.array_equal(y, y21[lo:hi]), (lo, hi)
        h.set_option("head_fuse", 0)
        y0, a0, l0 = h.forward(crops)
        assert not np.array_equal(l0, l21) and np.abs(l0 - l21).max() < 0.5


def test_head_conv_fused_with_pooling_f32(blob, taps, golden):
    """The same fusion for the parity configuration (head7.hip, v_mfma_f32_32x32x2_f32): against the oracle's pooled features at
    f32 accuracy, against the two-stage form (head_fuse=0: split-K GEMM, 49 x 1280 tensor written, pooled by the heads kernel),
    and bitwise batch invariance across the group sizes."""
    x = taps["b16/out"].astype(np.float32)
    want = taps["head"].mean(axis=(1, 2))
    with _lib.Handle(blob, device=0, dtype=_lib.F32) as h:
        r1 = h.op_head(x)
        h.set_option("head_fuse", 0)
        r0 = h.op_head(x)
        h.set_option("head_fuse", 1)
        assert not np.array_equal(r1["feat"], r0["feat"]), "head_fuse is not active"
        assert rel_err(r1["feat"], want) < 2e-6 and rel_err(r0["feat"], want) < 2e-6
        assert np.abs(r1["feat"] - r0["feat"]).max() < 2e-5
        assert np.abs(r1["logits"] - taps["logits"]).max() < 2e-3
        xs = np.concatenate([x] * 11)[:21]
        big = h.op_head(xs)
        for i in (0, 1, 15, 16, 17, 20):
            one = h.op_head(xs[i:i + 1])
            assert np.array_equal(big["feat"][i], one["feat"][0]) and np.array_equal(big["logits"][i], one["logits"][0]), i
        crops = np.concatenate([golden["crops"], synth.scene_crops(13, seed=78)])          # 21 crops
        y21, a21, l21 = h.forward(crops)
        for lo, hi in ((0, 1), (1, 3), (0, 16), (4, 21), (20, 21)):
            y, a, l = h.forward(crops[lo:hi])
            assert np.array_equal(l, l21[lo:hi]) and np.array_equal(y, y21[lo:hi]), (lo, hi)
        h.set_option("head_fuse", 0)
        y0, a0, l0 = h.forward(crops)
        assert np.abs(l0 - l21).max() < 1e-3 and np.abs(y0 - y21).max() < 1e-3


def test_stem_fused_with_block_1_depthwise_is_bitwise_the_two_kernels(blob, golden):
    """Round 4 (stemdw.hip; whenet.py:8, 23-26): for handles fed uint8 crops the stem conv is computed inside block 1's
    depthwise kernel -- the 112 x 112 x 32 stem output never reaches HBM.  The arithmetic is the two kernels' instruction for
    instruction: logits, angles and bins must be BITWISE those of option stem_fuse=0, for every batch split, with and
    without fold12 (f16) and for the f32 instantiation; one launch less; the float32-input entry point (no byte LUT) keeps
    the two kernels."""
    crops = np.concatenate([golden["crops"], synth.scene_crops(13, seed=31)])          # 21 crops
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        k1 = h.info().n_kernels_per_forward
        y1, a1, l1 = h.forward(crops)
        h.set_option("stem_fuse", 0)
        assert h.info().n_kernels_per_forward == k1 + 1
        y0, a0, l0 = h.forward(crops)
        assert np.array_equal(l1, l0) and np.array_equal(y1, y0) and np.array_equal(a1, a0)
        h.set_option("stem_fuse", 1)
        for lo, hi in ((0, 1), (1, 3), (0, 16), (4, 21), (20, 21)):
            y, a, l = h.forward(crops[lo:hi])
            assert np.array_equal(l, l1[lo:hi]) and np.array_equal(y, y1[lo:hi]), (lo, hi)
        h.set_option("fold12", 0)
        try:
            yf1, af1, lf1 = h.forward(crops)
            h.set_option("stem_fuse", 0)
            yf0, af0, lf0 = h.forward(crops)
            assert np.array_equal(lf1, lf0) and np.array_equal(yf1, yf0)
        finally:
            h.set_option("fold12", 1)
            h.set_option("stem_fuse", 1)
        d = h.device_alloc(crops[:8].nbytes)
        try:
            h.h2d(d, crops[:8])
            names = [s["kernel"] for s in h.profile(d, 8, 2)]
        finally:
            h.device_free(d)
        assert "whenet_stemdw_kernel" in names and not any("whenet_dw_kernel" in k or "whenet_stem_mfma" in k for k in names)
    with _lib.Handle(blob, device=0, dtype=_lib.F32) as h:           # the f32 instantiation (exact-f32 matrix-core stem)
        k1 = h.info().n_kernels_per_forward
        y1, a1, l1 = h.forward(crops)
        h.set_option("stem_fuse", 0)
        assert h.info().n_kernels_per_forward == k1 + 1
        y0, a0, l0 = h.forward(crops)
        assert np.array_equal(l1, l0) and np.array_equal(y1, y0) and np.array_equal(a1, a0)
        h.set_option("stem_fuse", 1)
        for lo, hi in ((0, 1), (1, 3), (4, 21)):
            y, a, l = h.forward(crops[lo:hi])
            assert np.array_equal(l, l1[lo:hi]) and np.array_equal(y, y1[lo:hi]), (lo, hi)
        # real-valued (already normalised) crops have no byte LUT: that entry point keeps the two kernels
        xf = O.normalise(crops[:3]).astype(np.float32)
        yf, af, lf = h.forward_f32(xf)
        assert np.abs(lf - l1[:3]).max() < 1e-3


def test_decode_kernel(handle):
    """utils.py:7-11 + whenet.py:28-33 on the device vs numpy float64, incl. ties and extremes."""
    rng = np.random.default_rng(5)
    lg = rng.normal(0, 5, size=(16, 252)).astype(np.float32)
    lg[0] = 0.0                         # all ties -> argmax 0, uniform softmax
    lg[1, :120] = -1e4
    lg[1, 7] = 50.0                     # one-hot
    lg[2, 100] = lg[2, 20] = 30.0       # exact tie: first index wins
    ypr, am = handle.op_decode(lg)
    y, p, r = O.decode(lg.astype(np.float64))
    assert np.abs(ypr - np.stack([y, p, r], 1)).max() < 2e-4
    assert np.array_equal(am, O.argmax_bins(lg))
    assert am[0].tolist() == [0, 0, 0] and am[1, 0] == 7 and am[2, 0] == 20


def test_end_to_end_golden(handle, golden):
    crops = golden["crops"]
    ypr, am, lg = handle.forward(crops)
    exp = golden["expected"]
    err = np.abs(ypr - exp["angles"]).max()
    print(f"\n[{handle.name}] max |angle - f64 oracle| = {err:.3e} deg; max |logit err| = "
          f"{np.abs(lg - exp['logits']).max():.3e}")
    if handle.name == "f32":
        assert err <= F32_DEG
        safe = exp["margins"] > MARGIN_F32
        assert safe.mean() > 0.9
        assert np.array_equal(am[safe], exp["argmax"][safe])
    else:
        assert err <= F16_DEG
        # bin argmax on every golden crop and head (no vacuous mask): either equal, or a flip to the
        # oracle's runner-up bin whose margin is below this run's logit error -- and only a few
        noise = float(np.abs(lg - exp["logits"]).max())
        flips = np.argwhere(am != exp["argmax"])
        print(f"[f16] argmax flips {len(flips)}/{am.size}; logit noise {noise:.3f}; "
              f"smallest oracle top-2 margin {exp['margins'].min():.3f}")
        assert len(flips) <= F16_ARGMAX_FLIPS, flips
        lo = {0: 0, 1: 120, 2: 186}
        nb = {0: 120, 1: 66, 2: 66}
        for i, h in flips:
            ref = exp["logits"][i, lo[h]:lo[h] + nb[h]]
            runner_up = int(np.argsort(ref)[-2])
            assert exp["margins"][i, h] <= 2 * noise, (i, h, exp["margins"][i, h], noise)
            assert am[i, h] == runner_up, (i, h, am[i, h], runner_up)
        big = exp["margins"] > 2 * noise
        assert np.array_equal(am[big], exp["argmax"][big])


def test_mfma_against_scalar_check_kernels(handle, golden):
    """The MFMA GEMMs vs the independent scalar-FMA kernels (same operands): validates the
    fragment / accumulator mapping on the device."""
    crops = golden["crops"][:3]
    _, _, lg0 = handle.forward(crops)
    handle.set_option("pw_impl", 1)
    try:
        _, _, lg1 = handle.forward(crops)
    finally:
        handle.set_option("pw_impl", 0)
    assert np.abs(lg0 - lg1).max() < (2e-3 if handle.name == "f32" else 0.5)


def test_batch_invariance_and_permutation(handle, golden):
    """Per-crop work is independent and every layer's launch configuration (hence its summation
    order) is fixed by the layer, never by the batch: a crop's result is BITWISE the same
    whatever the batch size, its position in the batch, or how the engine splits the batch over
    its concurrent streams."""
    crops = np.concatenate([golden["crops"], synth.scene_crops(5, seed=21)])       # 13: ragged
    ypr, am, lg = handle.forward(crops)
    for i in (0, 7, 12):
        y1, a1, l1 = handle.forward(crops[i:i + 1])
        assert np.array_equal(l1[0], lg[i]) and np.array_equal(y1[0], ypr[i]) and np.array_equal(a1[0], am[i])
    perm = np.random.default_rng(0).permutation(len(crops))
    yp, ap, lp = handle.forward(crops[perm])
    assert np.array_equal(lp, lg[perm]) and np.array_equal(yp, ypr[perm]) and np.array_equal(ap, am[perm])
    for n in (2, 3, 5, 9):
        y2, _, l2 = handle.forward(crops[:n])
        assert np.array_equal(l2, lg[:n])
    for lanes in (1, 2, 3):
        handle.set_option("lanes", lanes)
        handle.set_option("min_lane_crops", 2)
        try:
            y3, a3, l3 = handle.forward(crops)
        finally:
            handle.set_option("lanes", 2)              # the documented defaults (include/whenet_hip.h)
            handle.set_option("min_lane_crops", 16)
        assert np.array_equal(l3, lg) and np.array_equal(y3, ypr) and np.array_equal(a3, am)


def test_shard_sized_batches_are_bitwise_identical(handle):
    """What the multi-GPU batch shard relies on (BASELINE.json configs[3]: 64 crops per GPU):
    a 128-crop batch and its two 64-crop halves run separately give bitwise equal results."""
    crops = np.concatenate([synth.noise_crops(100, seed=4), synth.scene_crops(28, seed=5)])
    ypr, am, lg = handle.forward(crops)
    for lo in (0, 64):
        y, a, l = handle.forward(crops[lo:lo + 64])
        assert np.array_equal(l, lg[lo:lo + 64]) and np.array_equal(y, ypr[lo:lo + 64]) and np.array_equal(a, am[lo:lo + 64])


def test_graph_replay_equals_eager(handle, golden):
    crops = golden["crops"][:4]
    _, _, a = handle.forward(crops)
    _, _, b = handle.forward(crops)          # replay of the captured graph
    handle.set_option("graph", 0)
    try:
        _, _, c = handle.forward(crops)
    finally:
        handle.set_option("graph", 1)
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_submit_collect_pipeline(handle, golden):
    """Pinned-buffer submit/collect (up to 4 in flight) returns exactly what the blocking call
    returns for the same crops."""
    crops = golden["crops"]
    parts = ((0, 1), (1, 3), (4, 4))
    refs = [handle.forward(crops[i:i + k]) for i, k in parts]
    tickets = [handle.submit(crops[i:i + k]) for i, k in parts]
    for t, (i, k), (rypr, ram, rlg) in zip(tickets, parts, refs):
        ypr, am, lg = handle.collect(t, k, want_logits=True)
        assert np.array_equal(ypr, rypr) and np.array_equal(am, ram) and np.array_equal(lg, rlg)
    with pytest.raises(ValueError):
        handle.collect(12345, 1)


def test_full_size_batch_properties(handle, weights):
    """BASELINE.json configs[2] size (64 crops/GPU): spot-check 3 crops against the oracle and
    the whole batch through size-independent properties (determinism, permutation)."""
    crops = np.concatenate([synth.noise_crops(40, seed=0), synth.scene_crops(24, seed=3)])
    ypr, am, lg = handle.forward(crops)
    assert np.isfinite(lg).all() and np.isfinite(ypr).all()
    assert ypr[:, 0].min() >= -180 and ypr[:, 0].max() <= 177 and ypr[:, 1:].min() >= -99 and ypr[:, 1:].max() <= 96
    ypr2, am2, lg2 = handle.forward(crops)
    assert np.array_equal(lg, lg2)
    idx = [41, 50, 63]
    ref = O.forward(crops[idx], weights, np.float64)
    ang = np.stack([ref["yaw"], ref["pitch"], ref["roll"]], 1)
    assert np.abs(ypr[idx] - ang).max() <= (F32_DEG if handle.name == "f32" else F16_DEG)
    rev = crops[::-1].copy()
    _, _, lr = handle.forward(rev)
    assert np.array_equal(lr[::-1], lg)


def test_dropin_class_on_gpu(blob, golden, capsys):
    """The reference's call sequence (demo.py:20-22,14) against the drop-in module."""
    from whenet import WHENet
    m = WHENet(blob)
    assert m.model.summary() is None
    assert "Total params: 4,372,376" in capsys.readouterr().out
    crop = golden["crops"][:1]
    yaw, pitch, roll = m.get_angle(crop)
    for a in (yaw, pitch, roll):
        assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.shape == (1,)
    exp = golden["expected"]["angles"][0]
    assert abs(yaw[0] - exp[0]) < F32_DEG and abs(pitch[0] - exp[1]) < F32_DEG and abs(roll[0] - exp[2]) < F32_DEG
    # np.squeeze([yaw, pitch, roll]) as demo_video.py:28 does
    assert np.squeeze([yaw, pitch, roll]).shape == (3,)
    # Model.predict on the normalised image (whenet.py:23-27) returns the three logit arrays
    x = (crop / 255 - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    ly, lp, lr = m.model.predict(x, batch_size=8)
    assert ly.shape == (1, 120) and lp.shape == (1, 66) and lr.shape == (1, 66)
    assert np.array_equal(np.concatenate([ly, lp, lr], 1), m.last_logits)
    with pytest.raises(ValueError):
        m.get_angle(np.zeros((224, 224, 3), np.uint8))
    e = m.get_angle(np.zeros((0, 224, 224, 3), np.uint8))
    assert all(a.shape == (0,) for a in e)
    # input is not mutated, any integer-valued dtype works (the reference divides by 255)
    c2 = crop.astype(np.float64)
    y2, _, _ = m.get_angle(c2)
    assert np.array_equal(y2, yaw) and np.array_equal(c2, crop.astype(np.float64))
    m.close()


def test_real_valued_crops_follow_the_reference_arithmetic(blob, weights, golden):
    """whenet.py:25 divides ANY numeric array by 255 (float64), whenet.py:26 normalises, Keras casts
    to float32: non-integer crops take whenet_forward_f32 (no byte LUT) and must match the oracle fed
    the same float image; byte-valued crops through that path equal the LUT path to f32 round-off;
    Model.predict accepts any normalised float image."""
    from whenet import WHENet
    m = WHENet(blob)
    crops = golden["crops"][:3]
    soft = crops.astype(np.float64) * 0.7 + 11.3 + np.random.default_rng(2).uniform(0, 1, crops.shape)   # not integers
    y, p, r = m.get_angle(soft)
    x = ((soft / 255) - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    lg_ref = O.heads(O.backbone(x.astype(np.float32).astype(np.float64), weights), weights)
    yr, pr, rr = O.decode(lg_ref)
    assert np.abs(np.stack([y, p, r], 1) - np.stack([yr, pr, rr], 1)).max() <= F32_DEG
    assert np.abs(m.last_logits - lg_ref).max() < 2e-3
    # the float path on byte-valued input == the LUT path (same float32 network input)
    yb, pb, rb = m.get_angle(crops)
    lut_logits = m.last_logits.copy()
    xb = ((crops / 255) - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    ly, lp, lr = m.model.predict(xb)
    assert np.array_equal(np.concatenate([ly, lp, lr], 1), lut_logits)
    # Model.predict on an arbitrary float image (not the image of any byte crop)
    ly, lp, lr = m.model.predict(x + 0.01)
    ref2 = O.heads(O.backbone((x + 0.01).astype(np.float32).astype(np.float64), weights), weights)
    assert np.abs(np.concatenate([ly, lp, lr], 1) - ref2).max() < 2e-3
    with pytest.raises(ValueError):
        m.get_angle(np.array([["a"]], dtype=object).reshape(1, 1, 1, 1))
    m.close()


def test_hip_path_against_huggingface_fixture(handle, weights, golden):
    """Pin to a third-party implementation on the GPU box too: the committed 7x7x1280 features that
    HuggingFace transformers' EfficientNet computed from the same snapshot
    (tests/golden/hf_features.npz) -> GAP + Dense in numpy float64 -> logits; the HIP path's logits
    for those crops must match them (and so must the oracle's)."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "hf_features.npz"))
    crops = golden["crops"][fx["crop_index"]]
    lg_hf = O.heads(fx["features"].astype(np.float64), weights)
    assert np.abs(lg_hf - golden["expected"]["logits"][fx["crop_index"]]).max() < 1e-4      # oracle == HF
    ypr, am, lg = handle.forward(crops)
    y, p, r = O.decode(lg_hf)
    err = np.abs(ypr - np.stack([y, p, r], 1)).max()
    assert err <= (F32_DEG if handle.name == "f32" else F16_DEG), err
    if handle.name == "f32":
        assert np.abs(lg - lg_hf).max() < 2e-3
        assert np.array_equal(am, O.argmax_bins(lg_hf))


def test_keras_h5_snapshot_on_the_gpu(weights, golden, tmp_path):
    """whenet.py:15-16 / demo.py:20: WHENet('WHENet.h5').  The file is written in the Keras-2.1.6
    HDF5 layout by the h5py helper interpreter -- with the three Dense heads stored in another order
    than they were created (name matching) and layer numbering as if the model was built second in a
    session -- loaded through the drop-in constructor, and must reproduce the golden angles."""
    from whenet import WHENet
    from whenet_hip import keras_h5
    from tests.test_keras_h5 import shuffled_heads
    if not os.path.exists(keras_h5.HELPER):
        try:
            import h5py  # noqa: F401
        except ImportError:
            pytest.skip("no interpreter with h5py on this box")
    h5 = str(tmp_path / "WHENet.h5")
    keras_h5.write_keras_h5(h5, shuffled_heads(keras_h5.to_keras_layers(weights, offset=82)))
    m = WHENet(h5)
    y, p, r = m.get_angle(golden["crops"])
    assert np.abs(np.stack([y, p, r], 1) - golden["expected"]["angles"]).max() <= F32_DEG
    assert np.array_equal(m.last_argmax, golden["expected"]["argmax"])
    m.close()
    with pytest.raises(OSError):
        WHENet(str(tmp_path / "missing.h5"))
    junk = tmp_path / "junk.h5"
    junk.write_bytes(b"not hdf5 at all")
    with pytest.raises(ValueError):
        WHENet(str(junk))


def test_sharded_whenet_over_rccl_one_rank(blob, golden):
    """ShardedWHENet's REAL forward branch (libwhenet_hip handle on this rank's GPU, snapshot
    broadcast + result all-gather over the `nccl` = RCCL backend) with a 1-rank process group: what
    one GPU can execute of BASELINE.json configs[3].  Validation of the input happens on this path too."""
    import socket
    import torch
    import torch.distributed as dist
    from whenet_hip.shard import ShardedWHENet, broadcast_bytes
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert broadcast_bytes(blob[:4096], 0, dev) == blob[:4096]
        m = ShardedWHENet(blob, dtype="f32", device=0, comm_device=dev)
        assert m.world == 1 and m._model is not None
        y, p, r = m.get_angle(golden["crops"])
        assert np.abs(np.stack([y, p, r], 1) - golden["expected"]["angles"]).max() <= F32_DEG
        ypr, am = m.forward_local(golden["crops"][2:5], global_batch=False)
        assert np.array_equal(ypr[:, 0], y[2:5]) and am.shape == (3, 3)
        e, _ = m.forward_local(golden["crops"][:0], global_batch=False)
        assert e.shape == (0, 3)
        with pytest.raises(ValueError):
            m.get_angle(np.zeros((2, 100, 100, 3), np.uint8))          # would have read out of bounds
        with pytest.raises(ValueError):
            m.get_angle(np.zeros((2, 224, 224, 3), np.float32) + 0.5)  # bytes reinterpretation
        # a float array holding byte values is converted, not reinterpreted
        yf, _, _ = m.get_angle(golden["crops"].astype(np.float32))
        assert np.array_equal(yf, y)
        m.close()
    finally:
        dist.destroy_process_group()


def test_snapshot_file_roundtrip(weights_file, golden):
    from whenet import WHENet
    m = WHENet(snapshot=weights_file)
    y, p, r = m.get_angle(golden["crops"][:2])
    assert np.abs(np.stack([y, p, r], 1) - golden["expected"]["angles"][:2]).max() < F32_DEG
    m.close()


@pytest.mark.gpu
def test_forwards_in_flight_are_bitwise_identical(blob, golden):
    """option inflight: n engines behind one handle, used round-robin; every engine must produce the
    bits of the single-engine handle, through the device-pointer form and the pinned pipeline."""
    import torch
    crops = golden["crops"]
    n = crops.shape[0]
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        ref = h.forward(crops, want_logits=True)
        h.set_option("inflight", 3)
        assert h.forward(crops, want_logits=True)[0].tobytes() == ref[0].tobytes()
        dev = torch.device("cuda:0")
        d_crops = torch.from_numpy(crops).to(dev)
        outs = [(torch.zeros((n, 3), device=dev), torch.zeros((n, 3), dtype=torch.int32, device=dev),
                 torch.zeros((n, 252), device=dev)) for _ in range(6)]
        for y, a, l in outs:                                  # 6 forwards over 3 engines, all in flight
            h.forward_device(d_crops.data_ptr(), n, y.data_ptr(), a.data_ptr(), l.data_ptr())
        h.sync()
        for y, a, l in outs:
            assert np.array_equal(y.cpu().numpy(), ref[0]) and np.array_equal(a.cpu().numpy(), ref[1])
            assert np.array_equal(l.cpu().numpy(), ref[2])
        tickets = [h.submit(crops[i:i + 3]) for i in range(0, 6, 3)] + [h.submit(crops)]
        got = [h.collect(t, k, want_logits=True) for t, k in zip(tickets, (3, 3, n))]
        assert np.array_equal(got[0][0], ref[0][:3]) and np.array_equal(got[1][0], ref[0][3:6])
        assert np.array_equal(got[2][2], ref[2])
        with pytest.raises(ValueError):
            h.set_option("inflight", 9)
        h.set_option("inflight", 1)
        assert h.forward(crops)[0].tobytes() == ref[0].tobytes()


@pytest.mark.gpu
def test_large_batch_gemm_path_is_bitwise_the_small_batch_path(handle):
    """K >= 320 convs take the 2x2 register-blocked split-K kernel from 4096 rows (14x14 maps from 21
    crops per launch, 7x7 maps from 84): same partial sums, same bits as the crops run in chunks."""
    handle.set_option("lanes", 1)
    try:
        crops = np.concatenate([synth.scene_crops(40, seed=5), synth.noise_crops(56, seed=6)])     # 96 crops
        big = handle.forward(crops, want_logits=True)
        for i in range(0, 96, 16):
            small = handle.forward(crops[i:i + 16], want_logits=True)
            assert np.array_equal(small[2], big[2][i:i + 16]) and np.array_equal(small[0], big[0][i:i + 16])
    finally:
        handle.set_option("lanes", 2)


# ---------------------------------------------------------------------------------------------------------------
# round 3: every north-star batch size is a tested configuration (BASELINE.json: batch 1 / 8 / 64 / 512)
# ---------------------------------------------------------------------------------------------------------------
def _oracle_angles(crops, weights):
    ref = O.forward(crops, weights, np.float64)
    return np.stack([ref["yaw"], ref["pitch"], ref["roll"]], 1), ref


@pytest.mark.parametrize("schedule", ["default", "lanes3", "inflight3", "lanes1"])
def test_batch_512_bitwise_the_batch_64_path_and_against_the_oracle(handle, weights, schedule):
    """whenet.py:27 accepts any N.  512 crops (BASELINE.json configs[3]'s per-node batch; the project GEMMs take their
    NT = 2 / 3 tile instantiations here and nowhere below ~84 / 168 / 335 crops per launch) through the default
    schedule (2 lanes), with 3 lanes, with 3 forwards in flight, and as one 512-crop chain: bitwise the same crops run as
    8 x 64, and 9 crops spread over all lanes within tolerance of the float64 oracle."""
    crops = np.concatenate([synth.scene_crops(200, seed=41), synth.noise_crops(312, seed=42)])
    assert crops.shape[0] == 512
    if schedule == "inflight3":
        handle.set_option("inflight", 3)
    if schedule == "lanes1":
        handle.set_option("lanes", 1)
    if schedule == "lanes3":
        handle.set_option("lanes", 3)
    try:
        ypr, am, lg = handle.forward(crops)
        if schedule == "inflight3":          # the other two engines of the handle produce the same bits
            for _ in range(2):
                y2, a2, l2 = handle.forward(crops)
                assert np.array_equal(l2, lg) and np.array_equal(y2, ypr) and np.array_equal(a2, am)
    finally:
        handle.set_option("inflight", 1)
        handle.set_option("lanes", 2)
    assert np.isfinite(lg).all()
    for lo in range(0, 512, 64):
        y, a, l = handle.forward(crops[lo:lo + 64])
        assert np.array_equal(l, lg[lo:lo + 64]) and np.array_equal(y, ypr[lo:lo + 64]) and np.array_equal(a, am[lo:lo + 64]), lo
    idx = [0, 63, 170, 171, 255, 256, 341, 342, 511]          # first / last crop of each of the 2 or 3 lanes and between
    ang, ref = _oracle_angles(crops[idx], weights)
    err = np.abs(ypr[idx] - ang).max()
    assert err <= (F32_DEG if handle.name == "f32" else F16_DEG), err
    if handle.name == "f32":
        assert np.abs(lg[idx] - ref["logits"]).max() < 2e-3


def test_batch_8_and_batch_1(handle, weights, golden):
    """BASELINE.json batch 8 and batch 1: against the oracle, and bitwise the crops' results inside a larger batch."""
    crops = np.concatenate([synth.scene_crops(5, seed=51), synth.noise_crops(3, seed=52)])
    ypr, am, lg = handle.forward(crops)
    ang, ref = _oracle_angles(crops, weights)
    assert np.abs(ypr - ang).max() <= (F32_DEG if handle.name == "f32" else F16_DEG)
    big = np.concatenate([golden["crops"], crops, golden["crops"][:3]])            # 19 crops
    yb, ab, lb = handle.forward(big)
    assert np.array_equal(lb[8:16], lg) and np.array_equal(yb[8:16], ypr) and np.array_equal(ab[8:16], am)
    for i in (0, 7):
        y1, a1, l1 = handle.forward(crops[i:i + 1])
        assert np.array_equal(l1[0], lg[i]) and np.array_equal(y1[0], ypr[i])


def test_fused_squeeze_excite_is_bitwise_the_separate_launch(handle, golden):
    """Option se_fuse (default on): blocks 2-16's project GEMMs compute the SE gate of their rows' crops in their
    prologue (se_device.h) instead of reading the gate a squeeze-excite launch wrote.  Same arithmetic in the same order:
    every logit is bitwise the 51-launch schedule's, for ragged batches over all lanes."""
    crops = np.concatenate([golden["crops"], synth.scene_crops(45, seed=61)])          # 53 crops: 2 lanes of 26/27 (3 x 17/18 with lanes=3)
    y1, a1, l1 = handle.forward(crops)                     # default: se_fuse = 1
    try:
        for mode in (0, 2):
            handle.set_option("se_fuse", mode)
            y0, a0, l0 = handle.forward(crops)
            assert np.array_equal(l1, l0) and np.array_equal(y1, y0) and np.array_equal(a1, a0), mode
        # Round 6, option se_fuse = 3: blocks 7-16 compute their gate on the matrix cores inside the LDS-staged project GEMM (pw.hip
        # GM = 3: binary16 excite kernel and r vectors for f16 handles, hi/lo pairs for f32s) -- another rounding path, not another
        # result; ten launches fewer; measured slower (profiles/r06/se_mfma_ab.txt), so not the default.  The exact-f32 configuration
        # has no such form and stays bitwise.
        handle.set_option("se_fuse", 3)
        k3 = handle.info().n_kernels_per_forward
        y3, a3, l3 = handle.forward(crops)
        for n in (1, 5, 21):
            assert np.array_equal(handle.forward(crops[:n])[2], l3[:n]), n          # batch-invariant on its own
    finally:
        handle.set_option("se_fuse", SE_FUSE_DEFAULT)
    if handle.name == "f32":
        assert np.array_equal(l3, l1)
    else:
        assert not np.array_equal(l3, l1), "se_fuse=3 did not change the schedule"
        assert np.abs(l3 - l1).max() < (0.25 if handle.name == "f16" else 2e-4)
        assert k3 == handle.info().n_kernels_per_forward - 10


def test_xcd_placement_is_a_relabelling(handle, golden):
    """Option xcd_map (round 6): which XCD a workgroup of the fused front / 7 x 7 / head-conv kernels lands on is a relabelling of
    workgroups -- every logit bitwise the same for every mask, on batches whose unit counts are and are not multiples of eight."""
    crops = np.concatenate([golden["crops"], synth.scene_crops(29, seed=41)])          # 37 crops
    want = {n: handle.forward(crops[:n]) for n in (1, 3, 8, 21, 37)}
    try:
        for mask, concurrent in ((0, 1), (1, 1), (2, 1), (4, 1), (7, 1), (7, 0)):
            handle.set_option("xcd_map", mask)
            handle.set_option("concurrent", concurrent)        # (1: grouped at every batch, as a handle with inflight > 1; 0: from 128 crops per launch)
            for n, w in want.items():
                got = handle.forward(crops[:n])
                assert all(np.array_equal(a, b) for a, b in zip(got, w)), (mask, concurrent, n)
        with pytest.raises(ValueError):
            handle.set_option("xcd_map", 8)
    finally:
        handle.set_option("xcd_map", 7)
        handle.set_option("concurrent", 0)


def test_front_impl_variants_end_to_end(blob, golden):
    """f16: the network with front.hip on every block (round 2's schedule), with front2.hip on every block, and the
    default per-layer choice -- all within the f16 tolerance of the oracle, each batch-invariant."""
    exp = golden["expected"]["angles"]
    crops = golden["crops"]
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        outs = {}
        for impl in (0, 1, 2):
            h.set_option("front_impl", impl)
            y, a, l = h.forward(crops)
            assert np.abs(y - exp).max() <= F16_DEG, (impl, np.abs(y - exp).max())
            y1, a1, l1 = h.forward(crops[5:6])
            assert np.array_equal(l1[0], l[5]), impl
            many = np.concatenate([crops] * 7)                                        # 56 crops: 2 lanes
            ym, am_, lm = h.forward(many)
            assert np.array_equal(lm, np.concatenate([l] * 7)), impl
            outs[impl] = l
        assert np.abs(outs[2] - outs[0]).max() < 0.5 and np.abs(outs[1] - outs[0]).max() < 0.5
        # the default schedule without the block-1 / block-2 fold: same tolerance, same invariance
        h.set_option("front_impl", 1)
        h.set_option("fold12", 0)
        y, a, l = h.forward(crops)
        assert np.abs(y - exp).max() <= F16_DEG and np.abs(l - outs[1]).max() < 0.5
        assert not np.array_equal(l, outs[1])
        assert np.array_equal(h.forward(crops[5:6])[2][0], l[5])
        h.set_option("fold12", 1)
        assert np.array_equal(h.forward(crops)[2], outs[1])
        with pytest.raises(ValueError):
            h.set_option("front_impl", 3)


def _flips_are_runner_ups(am, lg, fx_logits, fx_argmax):
    """every bin flip goes to a bin whose ORACLE logit is within twice THIS crop's logit error of the oracle's maximum (a
    near tie that the f16 noise may legitimately resolve the other way: usually the runner-up, for a bimodal head another
    mode)"""
    flips = np.argwhere(am != fx_argmax)
    lo = {0: 0, 1: 120, 2: 186}
    nb = {0: 120, 1: 66, 2: 66}
    for i, hd in flips:
        ref = fx_logits[i, lo[hd]:lo[hd] + nb[hd]]
        noise = float(np.abs(lg[i] - fx_logits[i]).max())
        assert ref.max() - ref[am[i, hd]] <= 2 * noise, (i, hd, noise, float(ref.max() - ref[am[i, hd]]))
    return flips


def test_f16_accuracy_contract(blob):
    """The f16 product's error against the float64 oracle on 48 seeded crops that are NOT the golden crops
    (tests/golden/f16_set_expected.npz, generated by tests/golden/make_f16_set.py from oracle/whenet_oracle.py;
    measured: round 2 max 0.63 / p95 0.23 deg, round 3's default schedule max 0.87 / p95 0.23 deg, 0-1 bin flips of
    144): mean <= 0.065 deg, p95 <= 0.25 deg, max <= 1.0 deg (= F16_DEG), at most 2 bin flips, each to a bin within
    twice the crop's own logit error of the oracle's maximum."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "f16_set_expected.npz"))
    crops = np.concatenate([synth.scene_crops(24, seed=5), synth.noise_crops(24, seed=6)])
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        ypr, am, lg = h.forward(crops)
    e = np.abs(ypr - fx["angles"])
    noise = float(np.abs(lg - fx["logits"]).max())
    flips = _flips_are_runner_ups(am, lg, fx["logits"], fx["argmax"])
    print(f"\n[f16, 48 crops] max {e.max():.4f} mean {e.mean():.5f} p95 {np.percentile(e, 95):.4f} deg; "
          f"{len(flips)} bin flips of {am.size}; max |logit err| {noise:.4f}")
    assert e.mean() <= 0.065 and np.percentile(e, 95) <= 0.25 and e.max() <= F16_DEG
    assert len(flips) <= 2, flips
    with _lib.Handle(blob, device=0, dtype=_lib.F32) as h:
        y32, a32, _ = h.forward(crops)
    assert np.abs(y32 - fx["angles"]).max() <= F32_DEG


def test_f16_error_distribution_512_crops(blob):
    """The f16 error as a distribution, pinned to the ORACLE: 512 seeded crops (1536 angles) against the float64
    oracle's angles / logits / argmax (tests/golden/f16_set512_expected.npz, make_f16_set.py 512), for the default
    schedule and without option fold12.  The f32 configuration is held to the north-star bar on the same 512 crops
    (<= 1e-3 deg, argmax equal wherever the oracle's top-2 margin exceeds float32 round-off).
    Bounds (measured, profiles/r04/f16_error_gpu.txt: default mean 0.048 / p95 0.21 / p99 0.47 / p99.9 0.79 / max 1.21 deg, 12 flips;
    fold12=0 mean 0.055 / p95 0.24 / p99 0.52 / p99.9 1.00 / max 1.39 deg, 10 flips): mean <= 0.065, p95 <= 0.28, p99 <= 0.60, p99.9 <= 1.1,
    max <= F16_DEG_TAIL = 1.5 deg; <= 24 bin flips of 1536, every one of them to a bin whose oracle logit is within
    twice that crop's own logit error of the oracle's maximum."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "f16_set512_expected.npz"))
    crops = np.concatenate([synth.scene_crops(256, seed=41), synth.noise_crops(256, seed=42)])
    with _lib.Handle(blob, device=0, dtype=_lib.F32) as h32:
        y32, a32, l32 = h32.forward(crops)
    e32 = np.abs(y32 - fx["angles"])
    safe = fx["margins"] > MARGIN_F32
    print(f"\n[f32, 512 crops vs oracle] max {e32.max():.2e} deg; max |logit err| {np.abs(l32 - fx['logits']).max():.2e}; "
          f"{int((~safe).sum())} heads under the {MARGIN_F32} margin")
    assert e32.max() <= F32_DEG and safe.mean() > 0.97          # (31 of 1536 heads are closer than float32 round-off)
    assert np.array_equal(a32[safe], fx["argmax"][safe])
    for fold in (1, 0):
        with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
            h.set_option("fold12", fold)
            y, a, l = h.forward(crops)
        e = np.abs(y - fx["angles"])
        flips = _flips_are_runner_ups(a, l, fx["logits"], fx["argmax"])
        print(f"[f16 fold12={fold}, 512 crops vs oracle] mean {e.mean():.4f} p95 {np.percentile(e, 95):.4f} "
              f"p99 {np.percentile(e, 99):.4f} p99.9 {np.percentile(e, 99.9):.4f} max {e.max():.4f} deg; "
              f"{len(flips)} bin flips of {a.size}")
        assert e.mean() <= 0.065 and np.percentile(e, 95) <= 0.28 and np.percentile(e, 99) <= 0.60
        assert np.percentile(e, 99.9) <= 1.1 and e.max() <= F16_DEG_TAIL
        assert len(flips) <= 24                              # 1.5 % of the bins (near ties of neighbouring bins)


def test_no_kernel_reads_what_the_forward_did_not_write(handle):
    """Debug option poison: every activation buffer (both block buffers, expanded / depthwise tensors, head conv output,
    squeeze-excite partial sums and gates) is filled with NaN bit patterns before the forward starts; the results must
    not change -- for the default schedule, one lane, and without the block-1 / block-2 fold."""
    crops = synth.noise_crops(64, seed=9)
    defaults = {"lanes": 2, "fold12": 1, "se_fuse": SE_FUSE_DEFAULT, "front_impl": 1}
    try:
        for opts in ({}, {"lanes": 1}, {"fold12": 0}, {"se_fuse": 0}, {"se_fuse": 1}, {"front_impl": 0}):
            for k, v in opts.items():
                handle.set_option(k, v)
            handle.set_option("poison", 0)
            clean = handle.forward(crops)
            handle.set_option("poison", 1)
            dirty = handle.forward(crops)
            assert np.isfinite(dirty[2]).all() and all(np.array_equal(a, b) for a, b in zip(clean, dirty)), opts
            for k in opts:
                handle.set_option(k, defaults[k])
    finally:
        handle.set_option("poison", 0)
        for k, v in defaults.items():
            handle.set_option(k, v)


def test_replica_engines_first_forward_under_load(blob):
    """Regression (round 3): a replica engine's per-crop ticket counters (heads kernel) used to be zeroed by a
    null-stream hipMemset that the engine's non-blocking stream did not wait for; with the other engines keeping the
    GPU busy it could land in the middle of the replica's FIRST heads kernel and leave counters off by one for the
    life of the handle -- a few of the last crops of a 512-crop batch then summed stale partial logits.  Here: fresh
    handles, 512-crop batches submitted back to back through the ticket pipeline so that the replicas' first
    forwards start while engine 0 is busy; every result must be the reference bits."""
    crops = synth.noise_crops(512, seed=0)
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        ref = h.forward(crops)[2]
    for attempt in range(3):
        with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
            h.forward(crops[:512])                            # engine 0 sized and warm; replicas are created cold
            h.set_option("inflight", 3)
            for rnd in range(3):
                tickets = [h.submit(crops) for _ in range(3)]
                for t in tickets:
                    y, a, l = h.collect(t, 512, want_logits=True)
                    assert np.array_equal(l, ref), (attempt, rnd)


def test_bench_distributed_path_one_rank():
    """Keeps the 8-GPU path warm without an 8-GPU node: bench.py under WHENET_FORCE_DIST=1 runs the RCCL branch
    (process group, snapshot broadcast, barrier + max-over-ranks timing, per-rank all-gather) with one rank, weak and
    strong scaling; the JSON line must carry what the driver reads.  No scaling curve is claimed."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, WHENET_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
            "--no-latency", "--no-sweep", "--profile-iters", "2"]
    for extra, scaling, gb in (([], "weak", 64), (["--strong", "--global-batch", "512"], "strong", 512)):
        r = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["config"]["backend"] == "nccl (RCCL)" and line["config"]["world_size"] == 1
        assert line["scaling"] == scaling and line["config"]["global_batch"] == gb
        assert len(line["config"]["per_rank_crops_s"]) == 1 and line["value"] > 0 and line["n_gpus"] == 1
        assert line["steps"] == 5 and line["warmup"] == 2 and "roofline" in line
