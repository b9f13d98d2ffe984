"""Drop-in for the reference's detector POST-processing, on the GPU.

`yolo_eval` keeps the name, argument order and defaults of /root/reference/yolo_v3/model.py:193-199; the
reference builds TensorFlow graph ops from symbolic tensors and runs them in `sess.run`
(yolo_postprocess.py:102-104, 198-204), here the arguments are the numpy output maps of the detector
(`sess.run(yolo_model.output)`) and the result is numpy: boxes [k,4] (y_min, x_min, y_max, x_max), scores
[k], classes [k] -- what `YOLO.detect` returns (yolo_postprocess.py:205).  The detector itself stays where
it is (its weights are absent from the reference); there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib, weights as W

_default_handle: Optional[_lib.Handle] = None


def _handle() -> _lib.Handle:
    global _default_handle
    if _default_handle is None:
        # any handle provides the device, stream and scratch; the post-processing does not touch its weights
        _default_handle = _lib.Handle(W.pack(W.synthetic(1234)), device=0, dtype=_lib.F16)
    return _default_handle


def yolo_eval(yolo_outputs, anchors, num_classes, image_shape, max_boxes=20, score_threshold=.6, iou_threshold=.5,
              handle: Optional[_lib.Handle] = None):
    """Evaluate YOLO model on given input and return filtered boxes (model.py:193-232)."""
    h = handle if handle is not None else _handle()
    return h.yolo_eval(yolo_outputs, np.asarray(anchors, np.float32), int(num_classes), image_shape,
                       max_boxes=max_boxes, score_threshold=score_threshold, iou_threshold=iou_threshold)
