"""CPU restatement of the reference's per-head PRE-processing.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's checker
legs; the product (headposeestimation-whenet_amd/) never imports it.

**Parity: window arithmetic pinned to executed reference code, cv2.resize unpinned.**  The reference does the pixel
work with `cv2` (`opencv-python`, unpinned in /root/reference/requirements.txt:2), which is not installable here, and
it holds no test or recorded output for it.  The window arithmetic is the reference's own Python:
demo_video.process_detection is EXECUTED on 1,220 float32 boxes with a frame object that records the slice it is
asked for (tests/refharness.py -> tests/golden/reference_rects.npz) and crop_rect() must reproduce every window
(tests/test_reference_run.py).  What is restated:

* `enlarge_bbox`   /root/reference/demo_video.py:13-19  (the bbox margin arithmetic, float32
                   because YOLO's `sess.run` hands back float32 boxes -- yolo_postprocess.py:198-205
                   -- including its order dependence: y_max/x_max use the already-moved y_min/x_min)
* `crop`           /root/reference/demo_video.py:21 (`img[int(y_min):int(y_max), int(x_min):int(x_max)]`)
                   and /root/reference/demo.py:9-10 (integer bbox x_min,y_min,x_max,y_max)
* `bgr2rgb`        /root/reference/demo_video.py:22, demo.py:8 (`cv2.cvtColor(.., COLOR_BGR2RGB)`)
* `resize_linear_u8`  /root/reference/demo_video.py:23, demo.py:11 (`cv2.resize(img, (224, 224))`,
                   default INTER_LINEAR on 8-bit data).  OpenCV's published generic algorithm
                   (modules/imgproc/src/resize.cpp, unchanged from 2.4 to 4.x): coefficients
                   `fx = (float)((dx+0.5)*scale - 0.5)`, 11-bit fixed point
                   (`saturate_cast<short>(c * 2048)`), horizontal pass into int32 rows, vertical pass
                   `((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`; an exact 2x downscale is
                   switched to INTER_AREA (2x2 box mean, `(a+b+c+d+2)>>2`).  Vendor-optimised builds
                   (IPP) may differ from the generic path by one grey level; that cannot be
                   checked here.
"""
from __future__ import annotations

import numpy as np

OUT = 224
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def enlarge_bbox(frame_h: int, frame_w: int, bbox) -> tuple:
    """demo_video.py:13-19 on a YOLO box (y_min, x_min, y_max, x_max), float32 arithmetic.
    Returns the float32 window (y_min, x_min, y_max, x_max) after the margins."""
    f = np.float32
    y_min, x_min, y_max, x_max = (f(v) for v in bbox)
    y_min = max(0, y_min - abs(y_min - y_max) / f(10))
    y_max = min(frame_h, y_max + abs(y_min - y_max) / f(10))
    x_min = max(0, x_min - abs(x_min - x_max) / f(5))
    x_max = min(frame_w, x_max + abs(x_min - x_max) / f(5))
    x_max = min(x_max, frame_w)
    return y_min, x_min, y_max, x_max


def crop_rect(frame_h: int, frame_w: int, bbox) -> tuple:
    """The integer window demo_video.py:21 slices: (y0, x0, y1, x1) = int() of the enlarged box,
    clipped the way a numpy slice clips (stop > size -> size)."""
    y_min, x_min, y_max, x_max = enlarge_bbox(frame_h, frame_w, bbox)
    y0, x0, y1, x1 = int(y_min), int(x_min), int(y_max), int(x_max)
    return y0, x0, min(y1, frame_h), min(x1, frame_w)


def _round_half_even_to_short(v: np.ndarray) -> np.ndarray:
    """saturate_cast<short>(float) = cvRound (round half to even) + saturation."""
    return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int32)


def linear_tables(src: int, dst: int = OUT, horizontal: bool = True):
    """xofs / ialpha (horizontal) or yofs / ibeta (vertical) of OpenCV's INTER_LINEAR for one axis:
    returns (ofs[dst] int32, coef[dst,2] int32 (11-bit fixed point), nmax).  Horizontally the
    border samples are re-weighted (sx < 0 -> sample 0 with weight 1; from nmax on the pass reads
    the single sample ofs with weight 2048); vertically the tables are raw and the ROW INDICES are
    clamped when the rows are fetched."""
    scale = np.float64(src) / np.float64(dst)          # 1 / inv_scale, as resize() computes it
    ofs = np.zeros(dst, np.int32)
    coef = np.zeros((dst, 2), np.int32)
    nmax = dst
    for d in range(dst):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        fx = np.float32(fx - np.float32(sx))
        if horizontal:
            if sx < 0:
                fx, sx = np.float32(0), 0
            if sx + 1 >= src:
                nmax = min(nmax, d)
                if sx >= src - 1:
                    fx, sx = np.float32(0), src - 1
        ofs[d] = sx
        c = np.array([np.float32(1) - fx, fx], np.float32) * np.float32(COEF_SCALE)
        coef[d] = _round_half_even_to_short(c)
    return ofs, coef, nmax


def resize_linear_u8(src: np.ndarray, out: int = OUT) -> np.ndarray:
    """cv2.resize(src, (out, out)) for uint8 HxWxC, default interpolation (INTER_LINEAR)."""
    assert src.dtype == np.uint8 and src.ndim == 3
    h, w, _ = src.shape
    if h == 0 or w == 0:
        raise ValueError("empty crop (cv2.resize asserts !ssize.empty())")
    if h == 2 * out and w == 2 * out:                  # INTER_LINEAR -> INTER_AREA for an exact 2x shrink
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xofs, ialpha, xmax = linear_tables(w, out)
    yofs, ibeta, _ = linear_tables(h, out, horizontal=False)
    s = src.astype(np.int32)
    # horizontal pass: int32 rows, 11-bit fixed point
    x1 = np.minimum(xofs + 1, w - 1)
    rows = s[:, xofs, :] * ialpha[None, :, 0, None] + s[:, x1, :] * ialpha[None, :, 1, None]
    if xmax < out:
        rows[:, xmax:, :] = s[:, xofs[xmax:], :] * COEF_SCALE
    # vertical pass: rows sy and sy+1, clamped to the image (not re-weighted)
    y0 = np.clip(yofs, 0, h - 1)
    y1 = np.clip(yofs + 1, 0, h - 1)
    b0 = ibeta[:, 0, None, None]
    b1 = ibeta[:, 1, None, None]
    v = (((b0 * (rows[y0] >> 4)) >> 16) + ((b1 * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def crop_and_resize(frame: np.ndarray, rect, bgr2rgb: bool) -> np.ndarray:
    """One head of demo_video.py:21-23 (bgr2rgb=True) or demo.py:8-11 (frame already RGB)."""
    y0, x0, y1, x1 = rect
    c = frame[y0:y1, x0:x1]
    if bgr2rgb:
        c = c[:, :, ::-1]
    return resize_linear_u8(np.ascontiguousarray(c))


def frame_to_crops(frame_bgr: np.ndarray, bboxes) -> tuple:
    """All heads of a frame the way demo_video.py:56-58 visits them: returns (rects [k,4] int32,
    crops [k,224,224,3] uint8 RGB)."""
    h, w, _ = frame_bgr.shape
    rects = np.array([crop_rect(h, w, b) for b in bboxes], np.int32).reshape(-1, 4)
    crops = np.stack([crop_and_resize(frame_bgr, r, True) for r in rects]) if len(rects) else \
        np.zeros((0, OUT, OUT, 3), np.uint8)
    return rects, crops
