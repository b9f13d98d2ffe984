"""Parity tests proper: the HIP path, called through the C ABI, against the float64 oracle on
the same seeded inputs (pytest -m gpu, on the MI355X box).

Tolerances (stated here, per dtype):
  f32  angles <= 1e-3 deg (the north-star bar), bin argmax equal wherever the oracle's top-2
       logit margin exceeds 2e-3 (float32 round-off moves logits by ~3e-4, see
       tests/golden/golden.json noise_floor_*), per-kernel tensors <= 2e-5 * scale;
  f16  cannot meet 1e-3 deg (10-bit mantissa through 82 conv layers): angles <= F16_DEG,
       per-kernel tensors <= 4e-3 * scale (each kernel alone, fed oracle inputs rounded to f16).
"""
import os

import numpy as np
import pytest

from oracle import whenet_oracle as O
from whenet_hip import _lib, spec, synth, weights as W

pytestmark = pytest.mark.gpu

F32_DEG = 1e-3
F16_DEG = 1.0
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
    assert i.macs_per_crop == spec.TOTAL_MACS and i.n_kernels_per_forward == 51   # stem, dw(b1), 15 front, 16 se, 16 project, head conv, heads
    assert b"gfx950" in i.arch and i.compute_units >= 200


def test_stem_kernel(handle, taps):
    got = handle.op_stem(taps["crops"])
    assert rel_err(got, taps["stem"]) < tol(handle)


@pytest.mark.parametrize("index", list(range(1, 17)))
def test_mbconv_block_kernels(handle, taps, index):
    """expand GEMM, depthwise, SE gate, project GEMM (+skip) of every block, each on the
    oracle's own input for that block: covers every distinct layer shape of the network."""
    b = spec.blocks()[index - 1]
    x = taps["stem"] if index == 1 else taps[f"b{index - 1}/out"]
    t = tol(handle)
    p = f"b{index}"
    if b.has_expand:
        # the two-launch schedule materialises the expanded tensor: check it, then the default
        # (fused expand+depthwise, front.hip) must reproduce the same depthwise output bitwise
        handle.set_option("fuse_front", 0)
        try:
            r0 = handle.op_block(index, x.astype(np.float32))
        finally:
            handle.set_option("fuse_front", 1)
        assert rel_err(r0["expand"], taps[f"{p}/expand"]) < t, "expand"
        assert rel_err(r0["dw"], taps[f"{p}/dw"]) < 2 * t, "dw (unfused)"
        assert rel_err(r0["out"], taps[f"{p}/out"]) < 3 * t, "out (unfused)"
    r = handle.op_block(index, x.astype(np.float32))
    if b.has_expand:
        assert np.array_equal(r["dw"], r0["dw"]), "fused expand+depthwise differs from pw+dw"
        # the fused kernel also applies the SE reduce conv to its channel sums (another summation
        # order than se.hip's): same gate and block output within the kernel tolerance
        assert rel_err(r["gate"], r0["gate"]) < 2 * t and rel_err(r["out"], r0["out"]) < 3 * t
    # the stages below consume the kernel's own upstream output, so errors chain a little
    assert rel_err(r["dw"], taps[f"{p}/dw"]) < 2 * t, "dw"
    assert rel_err(r["gate"], taps[f"{p}/gate"].reshape(r["gate"].shape)) < 2 * t, "gate"
    assert rel_err(r["out"], taps[f"{p}/out"]) < 3 * t, "out"


def test_head_kernels(handle, taps):
    r = handle.op_head(taps["b16/out"].astype(np.float32))
    t = tol(handle)
    assert rel_err(r["feat"], taps["head"].mean(axis=(1, 2))) < 3 * t
    assert np.abs(r["logits"] - taps["logits"]).max() < (2e-3 if handle.name == "f32" else 1.0)


@pytest.mark.parametrize("nblk", list(range(1, 11)))
def test_tail_megakernel_block_chain(handle, taps, nblk):
    """The fused tail launch (blocks 7..16, one workgroup per crop), stopped after `nblk` blocks,
    on the oracle's block-6 output: covers every phase (expand GEMM -> LDS, depthwise from LDS,
    SE, gated project GEMM + skip) on every tail geometry (14x14 k3/k5, the stride-2 block 12,
    7x7 k5/k3)."""
    got = handle.op_tail(taps["b6/out"].astype(np.float32), nblk=nblk, dump=True)
    ref = taps[f"b{6 + nblk}/out"]
    assert got.shape == ref.shape
    assert rel_err(got, ref) < (1e-4 if handle.name == "f32" else 6e-2), nblk


def test_tail_megakernel_head(handle, taps):
    r = handle.op_tail(taps["b6/out"].astype(np.float32))
    assert rel_err(r["feat"], taps["head"].mean(axis=(1, 2))) < (1e-4 if handle.name == "f32" else 6e-2)
    assert np.abs(r["logits"] - taps["logits"]).max() < (2e-3 if handle.name == "f32" else 1.0)
    y, p, rr = O.decode(taps["logits"])
    assert np.abs(r["ypr"] - np.stack([y, p, rr], 1)).max() < (F32_DEG if handle.name == "f32" else F16_DEG)


def test_tail_fused_vs_layerwise(handle, golden):
    """Same network, two schedules: the fused tail launch and one launch per layer."""
    crops = golden["crops"]
    ypr0, am0, lg0 = handle.forward(crops)
    handle.set_option("tail", 1)
    try:
        ypr1, am1, lg1 = handle.forward(crops)
    finally:
        handle.set_option("tail", 0)
    assert np.abs(lg1 - lg0).max() < (2e-3 if handle.name == "f32" else 0.6)
    exp = golden["expected"]["angles"]
    for ypr in (ypr0, ypr1):
        assert np.abs(ypr - exp).max() <= (F32_DEG if handle.name == "f32" else F16_DEG)


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
        noise = np.abs(lg - exp["logits"]).max()
        safe = exp["margins"] > 4 * noise
        assert np.array_equal(am[safe], exp["argmax"][safe])


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
            handle.set_option("lanes", 4)
            handle.set_option("min_lane_crops", 8)
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
        handle.set_option("lanes", 3)
