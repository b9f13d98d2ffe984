"""Drop-in for the reference's detector POST-processing, on the GPU.

`yolo_eval` keeps the name, argument order and defaults of /root/reference/yolo_v3/model.py:193-199; the
reference builds TensorFlow graph ops from symbolic tensors and runs them in `sess.run`
(yolo_postprocess.py:102-104, 198-204), here the arguments are the numpy output maps of the detector
(`sess.run(yolo_model.output)`) and the result is numpy: boxes [k,4] (y_min, x_min, y_max, x_max), scores
[k], classes [k] -- what `YOLO.detect` returns (yolo_postprocess.py:205).  The detector itself stays where
it is (its weights are absent from the reference); there is no CPU fallback.
"""
from __future__ import annotations

import threading
from typing import Dict, Optional

import numpy as np

from . import _lib

_default_handles: Dict[int, _lib.Handle] = {}
_lock = threading.Lock()


def _handle(device: int) -> _lib.Handle:
    """One network-less handle per device (whenet_create_postproc: a device context, a stream and the scratch of
    the two post-processing launches -- no weights are synthesised or uploaded), created on first use."""
    with _lock:
        h = _default_handles.get(device)
        if h is None:
            h = _default_handles[device] = _lib.Handle.postproc(device)
        return h


def yolo_eval(yolo_outputs, anchors, num_classes, image_shape, max_boxes=20, score_threshold=.6, iou_threshold=.5,
              handle: Optional[_lib.Handle] = None, device: int = 0):
    """Evaluate YOLO model on given input and return filtered boxes (model.py:193-232).  `handle`: any whenet_hip
    handle (e.g. the WHENet model's) to run on; else a post-processing handle of `device` (handles are not
    thread-safe: callers that run this from several threads pass their own)."""
    h = handle if handle is not None else _handle(device)
    return h.yolo_eval(yolo_outputs, np.asarray(anchors, np.float32), int(num_classes), image_shape,
                       max_boxes=max_boxes, score_threshold=score_threshold, iou_threshold=iou_threshold)
