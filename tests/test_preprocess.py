"""Pre-processing rows: the oracle's restatement of demo_video.py:13-23 /
demo.py:8-11, the library's host arithmetic against it (CPU), and the crop/resize kernel and the
frame pipeline against it on the GPU (bit-exact: integer work)."""
import os

import numpy as np
import pytest

from oracle import preprocess_oracle as P
from whenet_hip import _lib, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sample_frames.npz")


# ------------------------------------------------------------------ oracle properties (CPU)
def test_resize_identity_constant_and_bilinear_bound():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_linear_u8(a), a)
    for hw in [(83, 64), (187, 164), (500, 300), (30, 20), (448, 448), (449, 448), (1, 1), (2, 3)]:
        c = np.full((*hw, 3), 137, np.uint8)
        assert (P.resize_linear_u8(c) == 137).all(), hw
    # within one grey level of real-valued half-pixel-centre bilinear interpolation
    src = rng.integers(0, 256, (37, 29, 3), dtype=np.uint8)
    h, w, _ = src.shape
    fy = (np.arange(224) + 0.5) * h / 224 - 0.5
    fx = (np.arange(224) + 0.5) * w / 224 - 0.5
    sy, sx = np.floor(fy).astype(int), np.floor(fx).astype(int)
    wy, wx = (fy - sy)[:, None, None], (fx - sx)[None, :, None]
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    x0, x1 = np.clip(sx, 0, w - 1), np.clip(sx + 1, 0, w - 1)
    s = src.astype(np.float64)
    ref = (1 - wy) * ((1 - wx) * s[y0][:, x0] + wx * s[y0][:, x1]) + wy * ((1 - wx) * s[y1][:, x0] + wx * s[y1][:, x1])
    assert np.abs(P.resize_linear_u8(src).astype(np.float64) - ref).max() < 1.0


def test_resize_exact_2x_is_box_mean():
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (448, 448, 3), dtype=np.uint8)
    s = src.astype(np.int32)
    want = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(P.resize_linear_u8(src), want.astype(np.uint8))


def test_bbox_margins_hand_worked_case():
    # one box worked by hand (float32: the second statement sees the moved y_min).  The reference's OWN
    # process_detection is executed on 1,220 boxes in tests/test_reference_run.py; this is only the readable example.
    f = np.float32
    y_min, x_min, y_max, x_max = f(100.5), f(200.25), f(300.75), f(380.5)
    e_ymin = y_min - abs(y_min - y_max) / f(10)
    e_ymax = y_max + abs(e_ymin - y_max) / f(10)
    e_xmin = x_min - abs(x_min - x_max) / f(5)
    e_xmax = x_max + abs(e_xmin - x_max) / f(5)
    got = P.enlarge_bbox(720, 1280, (100.5, 200.25, 300.75, 380.5))
    assert got == (e_ymin, e_xmin, e_ymax, e_xmax)
    assert e_ymax - y_max > y_min - e_ymin          # the order dependence: lower margin is larger
    assert P.crop_rect(720, 1280, (100.5, 200.25, 300.75, 380.5)) == (int(e_ymin), int(e_xmin), int(e_ymax), int(e_xmax))
    # clipping at the frame
    assert P.crop_rect(720, 1280, (2.0, 3.0, 150.0, 120.0))[:2] == (0, 0)
    assert P.crop_rect(720, 1280, (600.0, 1100.0, 719.0, 1279.0))[2:] == (720, 1280)


def test_library_rect_arithmetic_equals_oracle():
    """whenet_frame_rects is pure host code inside libwhenet_hip.so: checked on CPU."""
    rng = np.random.default_rng(3)
    for (h, w) in [(720, 1280), (224, 528), (1080, 1920), (97, 131)]:
        b = synth.head_boxes(200, h, w, seed=h)
        b = np.concatenate([b, np.array([[0, 0, h, w], [0.4, 0.6, 1.2, 1.4], [h - 3, w - 3, h, w]], np.float32)])
        b += rng.uniform(-3, 3, b.shape).astype(np.float32)        # some boxes stick out of the frame
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2])
        got = _lib.frame_rects(h, w, b)
        want = np.array([P.crop_rect(h, w, r) for r in b], np.int32)
        assert np.array_equal(got, want)
    assert _lib.frame_rects(10, 10, np.zeros((0, 4), np.float32)).shape == (0, 4)


def test_sample_fixture_matches_oracle():
    z = np.load(GOLD)
    for i in range(2):
        crop = P.crop_and_resize(z[f"frame{i}"], z[f"rect{i}"], bgr2rgb=True)
        assert np.array_equal(crop, z[f"crop{i}"])
    # Sample/bbox.txt: 240 0 304 83 / 116 0 280 187 (x_min y_min x_max y_max)
    assert z["rect0"].tolist() == [0, 240, 83, 304] and z["rect1"].tolist() == [0, 116, 187, 280]


# ------------------------------------------------------------------ GPU: kernel + pipeline
@pytest.fixture(scope="module")
def model():
    import whenet
    m = whenet.WHENet(dtype="f32")
    yield m
    m.close()


RECTS = [(80, 164, 322, 423), (0, 0, 165, 144), (588, 1064, 720, 1280), (0, 0, 720, 1280),
         (62, 36, 510, 484), (299, 639, 311, 652), (10, 20, 234, 244), (5, 5, 6, 6), (700, 0, 720, 1280),
         (0, 1279, 720, 1280), (100, 100, 101, 900), (3, 7, 5, 10)]


@pytest.mark.gpu
@pytest.mark.parametrize("bgr", [True, False])
def test_crop_resize_kernel_bit_exact(model, bgr):
    """every window class: interior, border-clipped, whole frame, 448x448 (INTER_AREA switch),
    tiny, identity 224x224, single pixel / row / column, strong up- and down-scaling"""
    frame = synth.video_frame()
    rects = np.array(RECTS, np.int32)
    assert (rects[4, 2] - rects[4, 0], rects[4, 3] - rects[4, 1]) == (448, 448)
    got = model._handle.op_crop_resize(frame, rects, bgr=bgr)
    for i, r in enumerate(rects):
        want = P.crop_and_resize(frame, r, bgr2rgb=bgr)
        assert np.array_equal(got[i], want), f"window {r.tolist()} differs"


@pytest.mark.gpu
def test_sample_images_through_the_kernel(model):
    z = np.load(GOLD)
    for i in range(2):
        got = model._handle.op_crop_resize(z[f"frame{i}"], z[f"rect{i}"][None], bgr=True)
        assert np.array_equal(got[0], z[f"crop{i}"])


@pytest.mark.gpu
def test_frame_pipeline_equals_per_head_get_angle(model):
    """demo_video.py:56-58 visits the heads one by one; the batched, device-cropped submission must
    give the same angles bitwise (crops are bit-exact, the forward is batch-invariant)."""
    from whenet_hip.frames import FramePipeline, crop_heads
    frame = synth.video_frame()
    boxes = synth.head_boxes(7)
    rects_o, crops_o = P.frame_to_crops(frame, boxes)
    rects_d, crops_d = crop_heads(model, frame, boxes)
    assert np.array_equal(rects_d, rects_o) and np.array_equal(crops_d, crops_o)
    per_head = [model.get_angle(c[None]) for c in crops_o]              # the reference's loop shape
    with FramePipeline(model, depth=2) as fp:
        rects, yaw, pitch, roll = fp.process(frame, boxes)
    assert np.array_equal(rects, rects_o)
    for i, (y, p, r) in enumerate(per_head):
        assert (yaw[i], pitch[i], roll[i]) == (y[0], p[0], r[0])


@pytest.mark.gpu
def test_frame_pipeline_order_depth_and_empty_frames(model):
    from whenet_hip.frames import FramePipeline
    frames = [synth.video_frame(360, 640, seed=s) for s in range(5)]
    boxes = [synth.head_boxes(k, 360, 640, seed=k) for k in (3, 0, 1, 5, 2)]
    want = []
    for f, b in zip(frames, boxes):
        _, crops = P.frame_to_crops(f, b)
        want.append(model.get_angle(crops))
    got = []
    with FramePipeline(model, depth=2) as fp:
        for f, b in zip(frames, boxes):
            if fp.in_flight == 2:
                got.append(fp.collect())
            fp.submit(f, b)
        with pytest.raises(ValueError):
            fp.submit(frames[0], boxes[0])                 # depth exceeded
        while fp.in_flight:
            got.append(fp.collect())
    assert len(got) == 5
    for (rects, yaw, pitch, roll), w, b in zip(got, want, boxes):
        assert rects.shape == (len(b), 4)
        assert np.array_equal(yaw, w[0]) and np.array_equal(pitch, w[1]) and np.array_equal(roll, w[2])


@pytest.mark.gpu
def test_frame_errors(model):
    frame = synth.video_frame(100, 100)
    with pytest.raises(ValueError):
        model._handle.op_crop_resize(frame, np.array([[10, 10, 10, 50]], np.int32))      # empty window
    with pytest.raises(ValueError):
        model._handle.op_crop_resize(frame, np.array([[0, 0, 101, 50]], np.int32))       # outside the frame
    with pytest.raises(ValueError):
        model._handle.submit_frame(frame[:, :, :2], np.array([[0, 0, 5, 5]], np.int32))   # not [H,W,3]
