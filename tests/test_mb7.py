"""Round 6, option mb7 (off by default: see DESIGN.md): blocks 13-16 of an f16 handle as ONE launch each (csrc/mb7.hip; reference: efficientnet 0.0.4 MBConvBlock + SEBlock as
instantiated by /root/reference/whenet.py:8) -- expand, depthwise, squeeze-excite, gate, project and skip in one workgroup per crop,
every intermediate tensor in LDS.  pytest -m gpu.

Tolerances: the per-kernel f16 tolerance of tests/test_gpu_parity.py (1.5e-2 * scale per stage fed oracle inputs; 2x / 3x for the
stages that consume the kernel's own upstream output) against the float64 oracle's taps, and the same against the three-launch
schedule of the same handle (front7.hip + se.hip + pw.hip, option mb7=0): the two differ by rounding points only (binary16
squeeze-excite kernels in mb7.hip, another summation order of the channel means and of the split-K partials)."""
import numpy as np
import pytest

from oracle import whenet_oracle as O
from whenet_hip import _lib, spec, synth, weights as W

pytestmark = pytest.mark.gpu
T16 = 1.5e-2


def rel_err(got, ref):
    ref = np.asarray(ref, np.float64)
    rms = max(np.sqrt((ref ** 2).mean()), 1e-6)
    return float((np.abs(np.asarray(got, np.float64) - ref) / (np.abs(ref) + rms)).max())


@pytest.fixture(scope="module")
def blob(weights):
    return W.pack(weights)


@pytest.fixture(scope="module")
def taps(weights, golden):
    crops = golden["crops"][[0, 3]]
    t = {}
    O.backbone(O.normalise(crops).astype(np.float64), weights, taps=t)
    return t


@pytest.fixture(scope="module")
def h16(blob):
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        h.set_option("mb7", 1)
        yield h


@pytest.mark.parametrize("index", [13, 14, 15, 16])
def test_one_launch_block_against_the_oracle_and_the_three_launch_schedule(h16, taps, index):
    x = taps[f"b{index - 1}/out"].astype(np.float32)
    b = spec.blocks()[index - 1]
    r1 = h16.op_block(index, x)                      # mb7.hip
    h16.set_option("mb7", 0)
    h16.set_option("se_fuse", 0)                     # (a squeeze-excite launch writes the gate)
    try:
        r3 = h16.op_block(index, x)
    finally:
        h16.set_option("se_fuse", 1)
        h16.set_option("mb7", 1)
    assert r1["out"].shape == (2, 7, 7, b.cout)
    assert not np.array_equal(r1["out"], r3["out"]), "mb7 is not active"
    # each stage against the oracle's tap of that stage
    assert rel_err(r1["dw"], taps[f"b{index}/dw"]) < 2 * T16, "dw"
    assert rel_err(r1["gate"], taps[f"b{index}/gate"].reshape(r1["gate"].shape)) < 2 * T16, "gate"
    assert rel_err(r1["out"], taps[f"b{index}/out"]) < 3 * T16, "out"
    # ... and against the three-launch schedule
    assert rel_err(r1["dw"], r3["dw"]) < T16 and rel_err(r1["gate"], r3["gate"]) < 2 * T16 and rel_err(r1["out"], r3["out"]) < 3 * T16
    assert np.isfinite(r1["out"]).all() and np.isfinite(r1["dw"]).all()


@pytest.mark.parametrize("index", [13, 16])
def test_one_launch_block_is_bitwise_independent_of_the_batch(h16, taps, index):
    """A crop's block output is bitwise what it is alone, wherever it sits in a ragged launch (one workgroup per crop, every order of
    summation fixed by the layer)."""
    x = taps[f"b{index - 1}/out"].astype(np.float32)
    rng = np.random.default_rng(5)
    xs = np.concatenate([x] * 11)[:21] * rng.uniform(0.5, 1.5, size=(21, 1, 1, 1)).astype(np.float32)
    big = h16.op_block(index, xs)
    for i in (0, 1, 7, 19, 20):
        one = h16.op_block(index, xs[i:i + 1])
        assert np.array_equal(big["out"][i], one["out"][0]) and np.array_equal(big["dw"][i], one["dw"][0]), (index, i)
    # the range operator (no debug taps written) gives the same bits as the single-block operator
    assert np.array_equal(h16.op_block_range(index, index, xs), big["out"])


def test_the_7x7_stage_as_a_range_against_the_oracle(h16, taps):
    x = taps["b12/out"].astype(np.float32)
    got = h16.op_block_range(13, 16, x)
    assert rel_err(got, taps["b16/out"]) < 8 * T16


def test_forward_with_the_one_launch_blocks(h16, golden):
    """Whole network: logits with mb7.hip on blocks 13-16 against the three-launch schedule -- another rounding path, not another
    result (the f16 distribution contract against the float64 oracle is tests/test_gpu_parity.py's); four kernels instead of twelve;
    bitwise independent of the batch split; the profile names the kernel."""
    crops = np.concatenate([golden["crops"], synth.scene_crops(13, seed=77)])          # 21 crops
    k1 = h16.info().n_kernels_per_forward
    y1, a1, l1 = h16.forward(crops)
    for lo, hi in ((0, 1), (1, 3), (3, 6), (0, 16), (4, 21), (20, 21)):
        y, a, l = h16.forward(crops[lo:hi])
        assert np.array_equal(l, l1[lo:hi]) and np.array_equal(y, y1[lo:hi]), (lo, hi)
    h16.set_option("mb7", 0)
    try:
        k0 = h16.info().n_kernels_per_forward
        y0, a0, l0 = h16.forward(crops)
    finally:
        h16.set_option("mb7", 1)
    assert k0 == k1 + 8
    assert not np.array_equal(l0, l1) and np.abs(l0 - l1).max() < 0.5 and np.abs(y0 - y1).max() < 1.0
    d = h16.device_alloc(crops[:8].nbytes)
    try:
        h16.h2d(d, crops[:8])
        names = [s["kernel"] for s in h16.profile(d, 8, 2)]
    finally:
        h16.device_free(d)
    assert sum("whenet_mb7_kernel" in k for k in names) == 4 and not any("front7" in k for k in names)


def test_f32_handles_keep_the_three_launch_schedule(blob):
    with _lib.Handle(blob, device=0, dtype=_lib.F32) as h:
        k = h.info().n_kernels_per_forward
        h.set_option("mb7", 1)
        assert h.info().n_kernels_per_forward == k
