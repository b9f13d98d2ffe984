"""Inputs of the reference-run fixtures (tests/golden/make_reference_fixtures.py): defined once, used by the
generator (build container, executes /root/reference) and by the tests that consume the fixtures (CPU and GPU)."""
import os

import numpy as np

from whenet_hip import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SIZES = (0, 1, 7, 8, 9, 64)                     # whenet.py:27 predicts in chunks of 8: below, at, above, many
FRAMES = ((720, 1280), (224, 528), (1080, 1920), (97, 131))
YOLO_CASES = (  # seed, classes, image (h, w), max_boxes, score, iou   (demo_video.py:74-75; YOLO._defaults; model.py:197-199)
    (1, 1, (720, 1280), 20, 0.3, 0.3),
    (3, 2, (720, 1280), 20, 0.3, 0.45),
    (4, 5, (1080, 607), 20, 0.6, 0.5),
    (7, 1, (720, 1280), 20, 0.3, 0.45, "tiny"),     # two output maps / six anchors: model.py:203's other anchor mask
    (9, 1, (1080, 607), 3, 0.3, 0.45, "rect"),      # a 10 x 13 grid (input 320 x 416), max_boxes below the candidate count
)


def yolo_case_maps(case):
    """The seeded detector maps and anchors of a YOLO_CASES entry."""
    seed, nc = case[0], case[1]
    kind = case[6] if len(case) > 6 else ""
    maps = synth.yolo_maps(seed, num_classes=nc)
    anchors = synth.YOLO_ANCHORS
    if kind == "tiny":
        maps, anchors = maps[:2], anchors[:6]
    elif kind == "rect":
        maps = [m[: m.shape[0] // 13 * 10] for m in maps]
    return maps, anchors


def crops64() -> np.ndarray:
    """The 8 committed golden crops followed by 56 seeded scene crops (whenet_hip/synth.py)."""
    g = np.load(os.path.join(GOLDEN, "golden_crops.npy"))
    return np.concatenate([g, synth.scene_crops(56, seed=2105)])


def real_valued(crops: np.ndarray) -> np.ndarray:
    """Non-byte input: fractional, some below 0 and above 255 -- the reference divides whatever it gets."""
    return crops[:3].astype(np.float64) * 1.0625 - 7.3


def all_bytes_image() -> np.ndarray:
    """One crop holding every byte value in every channel (rows 0-1), zeros elsewhere."""
    img = np.zeros((1, 224, 224, 3), np.uint8)
    v = np.arange(256, dtype=np.uint8)
    img[0, 0, :, :] = v[:224, None]
    img[0, 1, :32, :] = v[224:, None]
    return img


def lut_from_normalised(x: np.ndarray) -> np.ndarray:
    """[3,256] table out of the normalised all_bytes_image()."""
    return np.concatenate([x[0, 0, :, :], x[0, 1, :32, :]], axis=0).T.copy()


def boxes_for(h: int, w: int) -> np.ndarray:
    rng = np.random.default_rng(1000 + h)
    b = synth.head_boxes(300, h, w, seed=h)
    b = np.concatenate([b, np.array([[0, 0, h, w], [0.4, 0.6, 1.2, 1.4], [h - 3, w - 3, h, w],
                                     [2.0, 3.0, h / 5, w / 10], [h * 0.8, w * 0.85, h - 1, w - 1]], np.float32)])
    b = b + rng.uniform(-3, 3, b.shape).astype(np.float32)            # some boxes stick out of the frame
    b[:, 2:] = np.maximum(b[:, 2:], b[:, :2])
    return b.astype(np.float32)
