"""YOLOv3 post-processing: the numpy restatement on CPU, the HIP kernels against it on the GPU."""
import numpy as np
import pytest

from oracle import yolo_oracle as Y
from whenet_hip import synth

IMAGE = (720, 1280)          # demo_video.py feeds camera frames; letterboxed into 416x416


def test_oracle_known_answers():
    # one box decoded by hand: grid 1x1 (input 32x32), anchor (16, 8), zero logits, image = input
    feats = np.zeros((1, 1, 3 * 6), np.float32)
    b, s = Y.yolo_boxes_and_scores(feats, np.array([[16, 8], [16, 8], [16, 8]], np.float32), 1, (32, 32), (32, 32))
    assert b.shape == (3, 4) and s.shape == (3, 1)
    # centre (0.5, 0.5), w = 16/32, h = 8/32 -> y 12..20, x 8..24 pixels; score = 0.5 * 0.5
    assert np.allclose(b[0], [12, 8, 20, 24]) and np.allclose(s, 0.25)
    # letterbox: a 64x32 (h x w) image in a 32x32 input is scaled by 0.5 -> new shape 32x16, x offset 0.25
    b2, _ = Y.yolo_boxes_and_scores(feats, np.array([[16, 8]] * 3, np.float32), 1, (32, 32), (64, 32))
    assert np.allclose(b2[0], [24, 0, 40, 32])


def test_oracle_nms_semantics():
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10.5], [20, 20, 30, 30], [0, 0, 10, 30], [5, 5, 5, 9]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.95], np.float32)
    # box 4 has zero area: IoU 0 with everything, so it is kept (TensorFlow's IOU()); 1 is suppressed by 0
    assert Y.non_max_suppression(boxes, scores, 10, 0.5) == [4, 0, 2, 3]
    assert Y.non_max_suppression(boxes, scores, 2, 0.5) == [4, 0]
    # IoU exactly at the threshold is NOT suppressed (`>`): 10x10 vs 10x20 sharing the 10x10 -> IoU 0.5
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 20]], np.float32)
    assert Y.non_max_suppression(b, np.array([0.9, 0.8], np.float32), 10, 0.5) == [0, 1]
    # equal scores: lower index first
    assert Y.non_max_suppression(b, np.array([0.5, 0.5], np.float32), 1, 0.9) == [0]
    # corners given in the other order are normalised
    assert Y.iou(np.array([10, 10, 0, 0], np.float32), np.array([0, 0, 10, 10], np.float32)) == 1.0


def test_oracle_eval_structure():
    maps = synth.yolo_maps(3, num_classes=2)
    boxes, scores, classes, index = Y.yolo_eval(maps, synth.YOLO_ANCHORS, 2, IMAGE, max_boxes=20, score_threshold=0.3,
                                                iou_threshold=0.45, return_index=True)
    assert boxes.dtype == np.float32 and boxes.shape[1] == 4 and len(scores) == len(classes) == len(boxes) > 4
    assert list(classes) == sorted(classes)                                  # class by class
    for c in (0, 1):
        s = scores[classes == c]
        assert np.all(s[:-1] >= s[1:]) and np.all(s >= 0.3) and len(s) <= 20  # descending, thresholded, capped
        b = boxes[classes == c]
        for i in range(len(b)):
            for j in range(i):
                assert not Y.iou(b[i], b[j]) > np.float32(0.45)
    n_all = sum(m.shape[0] * m.shape[1] * 3 for m in maps)
    assert index.min() >= 0 and index.max() < n_all
    # the tiny configuration (2 maps, 6 anchors, model.py:203) uses the other anchor mask
    tb, ts, tc = Y.yolo_eval(maps[:2], synth.YOLO_ANCHORS[:6], 2, IMAGE, score_threshold=0.3, iou_threshold=0.45)
    assert len(tb) > 0


@pytest.fixture(scope="module")
def gpu_handle():
    from whenet_hip import _lib, weights as W
    h = _lib.Handle(W.pack(W.synthetic(1234)), device=0, dtype=_lib.F16)
    yield h
    h.close()


def _compare(h, maps, anchors, nc, image, **kw):
    rb, rs, rc, ri = Y.yolo_eval(maps, anchors, nc, image, return_index=True, **kw)
    gb, gs, gc, gi, all_boxes, all_scores = h.yolo_eval(maps, anchors, nc, image, debug=True, **kw)
    # every decoded box / score against the float32 restatement (expf vs numpy's exp: an ulp or two)
    ob, osc = [], []
    mask = [[6, 7, 8], [3, 4, 5], [0, 1, 2]] if len(maps) == 3 else [[3, 4, 5], [1, 2, 3]]
    inp = (maps[0].shape[0] * 32, maps[0].shape[1] * 32)
    for l, m in enumerate(maps):
        b, s = Y.yolo_boxes_and_scores(m, np.asarray(anchors, np.float32).reshape(-1, 2)[mask[l]], nc, inp, image)
        ob.append(b)
        osc.append(s)
    ob, osc = np.concatenate(ob), np.concatenate(osc)
    assert np.allclose(all_scores, osc, rtol=2e-6, atol=1e-7)
    assert np.allclose(all_boxes, ob, rtol=1e-5, atol=1e-3 * max(image))
    # the selection itself: same boxes in the same order
    assert list(gc) == list(rc) and list(gi) == list(ri), (gi, ri)
    assert np.allclose(gs, rs, rtol=2e-6) and np.allclose(gb, rb, rtol=1e-5, atol=1e-3 * max(image))
    return len(gb)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,nc", [(1, 1), (2, 1), (3, 2), (4, 5)])
def test_gpu_yolo_eval_matches_oracle(gpu_handle, seed, nc):
    maps = synth.yolo_maps(seed, num_classes=nc)
    # the reference's own settings: demo_video.py:74-75 (score 0.3, iou 0.3), YOLO defaults 0.3 / 0.45, model.py 0.6 / 0.5
    for score, iou in ((0.3, 0.3), (0.3, 0.45), (0.6, 0.5)):
        n = _compare(gpu_handle, maps, synth.YOLO_ANCHORS, nc, IMAGE, max_boxes=20, score_threshold=score, iou_threshold=iou)
        assert n > 0


@pytest.mark.gpu
def test_gpu_yolo_eval_edges(gpu_handle):
    maps = synth.yolo_maps(7, num_classes=1)
    # nothing passes the threshold: empty result
    b, s, c = gpu_handle.yolo_eval(maps, synth.YOLO_ANCHORS, 1, IMAGE, score_threshold=1.5)
    assert b.shape == (0, 4) and s.shape == (0,) and c.shape == (0,)
    # everything passes (threshold 0): 10,647 candidates sorted in global memory, capped at max_boxes
    assert _compare(gpu_handle, maps, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=20, score_threshold=0.0, iou_threshold=0.45) == 20
    assert _compare(gpu_handle, maps, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=256, score_threshold=0.05, iou_threshold=0.9) > 20
    # max_boxes = 1, the tiny configuration, a non-square grid, a portrait image
    assert _compare(gpu_handle, maps, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=1, score_threshold=0.3, iou_threshold=0.45) == 1
    _compare(gpu_handle, maps[:2], synth.YOLO_ANCHORS[:6], 1, IMAGE, max_boxes=20, score_threshold=0.3, iou_threshold=0.45)
    rect = [m[: m.shape[0] // 13 * 10] for m in synth.yolo_maps(9, num_classes=1)]       # 10x13, 20x26, 40x52
    _compare(gpu_handle, rect, synth.YOLO_ANCHORS, 1, (1080, 607), max_boxes=20, score_threshold=0.3, iou_threshold=0.45)
    # duplicated candidates (equal scores): the lower index wins
    dup = [m.copy() for m in maps]
    dup[0][5, 6] = dup[0][5, 5]
    _compare(gpu_handle, dup, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=20, score_threshold=0.1, iou_threshold=0.99)
    # argument errors
    with pytest.raises(ValueError):
        gpu_handle.yolo_eval(maps, synth.YOLO_ANCHORS[:6], 1, IMAGE)
    # any max_boxes, as the reference: more than 256 selections per class spill from LDS to the output array
    # (every box passes, nothing overlaps enough to be suppressed at IoU 0.999 -> hundreds of detections)
    assert _compare(gpu_handle, maps, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=1000, score_threshold=0.0, iou_threshold=0.999) > 256
    assert _compare(gpu_handle, maps, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=10 ** 7, score_threshold=0.3, iou_threshold=0.45) > 0
    with pytest.raises(ValueError):
        gpu_handle.yolo_eval(maps, synth.YOLO_ANCHORS, 1, IMAGE, max_boxes=0)
    with pytest.raises(ValueError):
        gpu_handle.yolo_eval(maps, synth.YOLO_ANCHORS, 2, IMAGE)


@pytest.mark.gpu
def test_gpu_yolo_drop_in_signature(gpu_handle):
    from whenet_hip import yolo
    maps = [m[None] for m in synth.yolo_maps(11, num_classes=1)]          # with the batch axis sess.run returns
    b, s, c = yolo.yolo_eval(maps, synth.YOLO_ANCHORS, 1, IMAGE, score_threshold=0.3, iou_threshold=0.45, handle=gpu_handle)
    rb, rs, rc = Y.yolo_eval([m[0] for m in maps], synth.YOLO_ANCHORS, 1, IMAGE, score_threshold=0.3, iou_threshold=0.45)
    assert b.shape == rb.shape and np.allclose(b, rb, rtol=1e-5, atol=1.0) and list(c) == list(rc)
    # without a handle: the module's network-less post-processing handle (whenet_create_postproc), same result
    b2, s2, c2 = yolo.yolo_eval(maps, synth.YOLO_ANCHORS, 1, IMAGE, score_threshold=0.3, iou_threshold=0.45)
    assert np.array_equal(b2, b) and np.array_equal(s2, s)
    with pytest.raises(ValueError):                               # such a handle has no network to run
        yolo._handle(0).forward(np.zeros((1, 224, 224, 3), np.uint8))
