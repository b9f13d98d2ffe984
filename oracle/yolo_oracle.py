"""CPU restatement of the reference's YOLOv3 POST-processing.

TEST INFRASTRUCTURE ONLY -- imported by tests/ and __graft_entry__.smoke(); the product
(headposeestimation-whenet_amd/) never imports it.

**Parity: decode pinned to executed reference code, NMS unpinned.**  The reference runs this as TensorFlow-1.12 graph
ops inside `sess.run` (/root/reference/yolo_v3/yolo_postprocess.py:198-204); TensorFlow cannot be installed here and
the reference records no detections.  But yolo_v3/model.py:125-232 is Python over `keras.backend`: it is EXECUTED over a
numpy float32 backend stand-in (tests/refharness.py) and its decoded boxes / scores / selections are committed as
tests/golden/reference_yolo.npz; tests/test_reference_run.py asserts this module reproduces them.  Only
`tf.image.non_max_suppression` (a C++ kernel) stays a restatement.  What is restated, in float32 like the graph:

* `yolo_head`              /root/reference/yolo_v3/model.py:125-150  (grid, sigmoid/exp decode, anchors)
* `yolo_correct_boxes`     /root/reference/yolo_v3/model.py:153-178  (letterbox offset/scale, y-first boxes,
                           `K.round` = round-half-to-even)
* `yolo_boxes_and_scores`  /root/reference/yolo_v3/model.py:181-190
* `yolo_eval`              /root/reference/yolo_v3/model.py:193-232  (anchor masks, score threshold `>=`,
                           per-class NMS, concatenation class by class)
* `non_max_suppression`    `tf.image.non_max_suppression` (TensorFlow 1.12,
                           core/kernels/non_max_suppression_op.cc): candidates taken in descending score
                           order, a candidate is dropped when its IoU with an already selected box is
                           `> iou_threshold`, at most `max_output_size` are kept; boxes are normalised with
                           min/max per axis, a box of non-positive area has IoU 0.  TensorFlow leaves the
                           order of EQUAL scores to its heap; here ties go to the lower box index.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

F = np.float32


def sigmoid(x: np.ndarray) -> np.ndarray:
    return (F(1) / (F(1) + np.exp(-x.astype(F)))).astype(F)


def yolo_head(feats: np.ndarray, anchors: np.ndarray, num_classes: int, input_shape: Sequence[int]):
    """model.py:125-150.  feats [gh, gw, A*(5+C)] -> box_xy, box_wh [gh,gw,A,2], conf [gh,gw,A,1], probs [gh,gw,A,C]."""
    na = len(anchors)
    gh, gw = feats.shape[0], feats.shape[1]
    f = feats.astype(F).reshape(gh, gw, na, num_classes + 5)
    grid_y = np.tile(np.arange(gh).reshape(-1, 1, 1, 1), [1, gw, 1, 1])
    grid_x = np.tile(np.arange(gw).reshape(1, -1, 1, 1), [gh, 1, 1, 1])
    grid = np.concatenate([grid_x, grid_y], axis=-1).astype(F)
    box_xy = ((sigmoid(f[..., :2]) + grid) / np.array([gw, gh], F)).astype(F)
    box_wh = (np.exp(f[..., 2:4]) * anchors.astype(F).reshape(1, 1, na, 2) /
              np.array([input_shape[1], input_shape[0]], F)).astype(F)
    return box_xy, box_wh, sigmoid(f[..., 4:5]), sigmoid(f[..., 5:])


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape) -> np.ndarray:
    """model.py:153-178 -> [...,4] = y_min, x_min, y_max, x_max in image pixels."""
    box_yx = box_xy[..., ::-1]
    box_hw = box_wh[..., ::-1]
    input_shape = np.asarray(input_shape, F)
    image_shape = np.asarray(image_shape, F)
    new_shape = np.round(image_shape * np.min(input_shape / image_shape)).astype(F)     # round half to even
    offset = ((input_shape - new_shape) / F(2.) / input_shape).astype(F)
    scale = (input_shape / new_shape).astype(F)
    box_yx = ((box_yx - offset) * scale).astype(F)
    box_hw = (box_hw * scale).astype(F)
    box_mins = box_yx - (box_hw / F(2.))
    box_maxes = box_yx + (box_hw / F(2.))
    boxes = np.concatenate([box_mins[..., 0:1], box_mins[..., 1:2], box_maxes[..., 0:1], box_maxes[..., 1:2]], axis=-1)
    return (boxes * np.concatenate([image_shape, image_shape])).astype(F)


def yolo_boxes_and_scores(feats, anchors, num_classes, input_shape, image_shape):
    """model.py:181-190 -> boxes [n,4], scores [n,C], n = gh*gw*A in (y, x, anchor) order."""
    box_xy, box_wh, conf, probs = yolo_head(feats, anchors, num_classes, input_shape)
    boxes = yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape).reshape(-1, 4)
    scores = (conf * probs).astype(F).reshape(-1, num_classes)
    return boxes, scores


def iou(a: np.ndarray, b: np.ndarray) -> np.float32:
    """TensorFlow's IOU() of non_max_suppression_op.cc, float32."""
    ymin_a, xmin_a = min(a[0], a[2]), min(a[1], a[3])
    ymax_a, xmax_a = max(a[0], a[2]), max(a[1], a[3])
    ymin_b, xmin_b = min(b[0], b[2]), min(b[1], b[3])
    ymax_b, xmax_b = max(b[0], b[2]), max(b[1], b[3])
    area_a = F(F(ymax_a - ymin_a) * F(xmax_a - xmin_a))
    area_b = F(F(ymax_b - ymin_b) * F(xmax_b - xmin_b))
    if area_a <= 0 or area_b <= 0:
        return F(0)
    iymin, ixmin = max(ymin_a, ymin_b), max(xmin_a, xmin_b)
    iymax, ixmax = min(ymax_a, ymax_b), min(xmax_a, xmax_b)
    inter = F(max(F(iymax - iymin), F(0)) * max(F(ixmax - ixmin), F(0)))
    return F(inter / F(F(area_a + area_b) - inter))


def non_max_suppression(boxes: np.ndarray, scores: np.ndarray, max_output_size: int, iou_threshold: float) -> List[int]:
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    thr = F(iou_threshold)
    keep: List[int] = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        if all(not (iou(boxes[i], boxes[j]) > thr) for j in keep):
            keep.append(i)
    return keep


def yolo_eval(yolo_outputs: Sequence[np.ndarray], anchors: np.ndarray, num_classes: int, image_shape,
              max_boxes: int = 20, score_threshold: float = .6, iou_threshold: float = .5,
              return_index: bool = False) -> Tuple[np.ndarray, ...]:
    """model.py:193-232 on numpy feature maps [gh, gw, A*(5+C)] (batch of one, as detect() feeds it).
    Returns boxes [k,4] (y_min, x_min, y_max, x_max), scores [k], classes [k] (int32), class by class."""
    num_layers = len(yolo_outputs)
    anchor_mask = [[6, 7, 8], [3, 4, 5], [0, 1, 2]] if num_layers == 3 else [[3, 4, 5], [1, 2, 3]]
    anchors = np.asarray(anchors, F).reshape(-1, 2)
    input_shape = (yolo_outputs[0].shape[0] * 32, yolo_outputs[0].shape[1] * 32)
    boxes, box_scores = [], []
    for l in range(num_layers):
        b, s = yolo_boxes_and_scores(yolo_outputs[l], anchors[anchor_mask[l]], num_classes, input_shape, image_shape)
        boxes.append(b)
        box_scores.append(s)
    boxes = np.concatenate(boxes, axis=0)
    box_scores = np.concatenate(box_scores, axis=0)
    mask = box_scores >= F(score_threshold)
    out_b, out_s, out_c, out_i = [], [], [], []
    for c in range(num_classes):
        idx = np.nonzero(mask[:, c])[0]
        keep = non_max_suppression(boxes[idx], box_scores[idx, c], max_boxes, iou_threshold)
        out_b.append(boxes[idx][keep].reshape(-1, 4))
        out_s.append(box_scores[idx, c][keep])
        out_c.append(np.full(len(keep), c, np.int32))
        out_i.append(idx[keep].astype(np.int32))
    res = (np.concatenate(out_b, axis=0).astype(F), np.concatenate(out_s).astype(F), np.concatenate(out_c))
    return res + (np.concatenate(out_i),) if return_index else res
