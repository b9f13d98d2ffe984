"""Per-frame driver for video callers: all heads of a frame in ONE submission, overlapped with the
caller's next detection.

The reference's video loop (/root/reference/demo_video.py:49-63) runs, per frame, YOLO on the host
and then `process_detection` once per head, sequentially: bbox margins (13-19), numpy slice (21),
cv2.cvtColor (22), cv2.resize to 224x224 (23) and a batch-1 `get_angle` (27).  Here

  * the margin arithmetic runs once per frame inside the library (`whenet_frame_rects`);
  * the frame crosses PCIe once (pinned staging copy + async H2D) and every head is cropped /
    BGR->RGB-swapped / resized by one kernel straight into the forward's input (`csrc/frame.hip`,
    bit-exact with OpenCV's generic fixed-point INTER_LINEAR as restated in the oracle);
  * the k heads go through the network as ONE batch;
  * `submit()` returns as soon as the work is enqueued, so the caller can run the detector on the
    next frame while this one is on the GPU; `collect()` returns results in submission order.

Only numpy and the C ABI are used (no torch, no cv2); there is no CPU fallback.
"""
from __future__ import annotations

from collections import deque
from typing import Deque, Tuple

import numpy as np

from . import _lib

MAX_INFLIGHT = 4          # WHENET_MAX_INFLIGHT


class FramePipeline:
    """`with FramePipeline(model) as fp:` ... `fp.submit(frame_bgr, bboxes)` ... `fp.collect()`.

    `model` is a `whenet.WHENet`; `bboxes` is what `YOLO.detect` returns first: float32 [k,4] rows
    (y_min, x_min, y_max, x_max) in frame pixels.  `collect()` returns
    `(rects, yaw, pitch, roll)`: rects int32 [k,4] = the enlarged windows (y0, x0, y1, x1) that
    demo_video.py:25,29 also needs for drawing, and three float32 (k,) arrays as `get_angle`
    returns them."""

    def __init__(self, model, depth: int = 2, bgr: bool = True):
        if not 1 <= depth <= MAX_INFLIGHT:
            raise ValueError(f"depth must be 1..{MAX_INFLIGHT}")
        self._h = model._handle
        # `depth` frames in flight = `depth` engines behind the handle (own streams / arena / graphs):
        # frame i+1's staging, crop and forward overlap frame i's on the GPU
        self._h.set_option("inflight", min(depth, 4))
        self._depth = depth
        self._bgr = bool(bgr)
        self._pending: Deque[Tuple[int, np.ndarray]] = deque()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        while self._pending:
            self.collect()

    @property
    def in_flight(self) -> int:
        return len(self._pending)

    def submit(self, frame: np.ndarray, bboxes) -> None:
        """Enqueue one frame; raises ValueError if `depth` frames are already in flight
        (collect one first) or if a box degenerates to an empty window (cv2.resize would raise)."""
        if len(self._pending) >= self._depth:
            raise ValueError(f"{self._depth} frames already in flight: collect() first")
        frame = np.asarray(frame)
        if frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8:
            raise ValueError(f"frame must be uint8 [H,W,3], got {frame.dtype} {frame.shape}")
        rects = _lib.frame_rects(frame.shape[0], frame.shape[1], bboxes)
        ticket = self._h.submit_frame(frame, rects, bgr=self._bgr)
        self._pending.append((ticket, rects))

    def collect(self):
        """Oldest submitted frame -> (rects [k,4] int32, yaw, pitch, roll float32 (k,))."""
        if not self._pending:
            raise ValueError("nothing in flight")
        ticket, rects = self._pending.popleft()
        k = rects.shape[0]
        ypr, _, _ = self._h.collect(ticket, k)
        return rects, ypr[:, 0].copy(), ypr[:, 1].copy(), ypr[:, 2].copy()

    def process(self, frame: np.ndarray, bboxes):
        """Synchronous form: one frame in, its heads' angles out."""
        self.submit(frame, bboxes)
        while len(self._pending) > 1:
            self.collect()
        return self.collect()


def crop_heads(model, frame: np.ndarray, bboxes, bgr: bool = True):
    """The crops `process_detection` would have handed to `get_angle`, made on the device:
    returns (rects int32 [k,4], crops uint8 [k,224,224,3] RGB)."""
    frame = np.asarray(frame)
    rects = _lib.frame_rects(frame.shape[0], frame.shape[1], bboxes)
    if rects.shape[0] == 0:
        return rects, np.zeros((0, 224, 224, 3), np.uint8)
    return rects, model._handle.op_crop_resize(frame, rects, bgr=bgr)
