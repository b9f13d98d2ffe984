"""Seeded synthetic 224x224 RGB uint8 head crops (there is no dataset in the reference and
no network): the inputs every parity test and bench line in this repo is quoted on.

  noise_crops   iid uniform bytes -- BASELINE.md §4 / SURVEY.md §8d: the throughput input.
  scene_crops   low-frequency colour fields + blobs + edges: crops whose deep features
                differ from one another the way real head crops do (iid noise crops all
                look alike to a CNN, which would make the parity tests insensitive).
"""
from __future__ import annotations

import numpy as np

IMG = 224


def noise_crops(n: int, seed: int = 0) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, (n, IMG, IMG, 3), dtype=np.uint8)


def scene_crops(n: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, IMG), np.linspace(-1, 1, IMG), indexing="ij")
    out = np.empty((n, IMG, IMG, 3), dtype=np.uint8)
    for i in range(n):
        img = np.zeros((IMG, IMG, 3))
        base = rng.uniform(0.1, 0.9, size=3)
        grad = rng.normal(0, 0.25, size=(2, 3))
        img += base + yy[..., None] * grad[0] + xx[..., None] * grad[1]
        for _ in range(int(rng.integers(2, 6))):            # plane waves
            f = rng.uniform(0.5, 9.0, size=2) * rng.choice([-1, 1], size=2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.normal(0, 0.15, size=3)
            img += np.cos(np.pi * (f[0] * yy + f[1] * xx) + ph)[..., None] * amp
        for _ in range(int(rng.integers(1, 5))):            # soft-edged ellipses ("heads")
            cy, cx = rng.uniform(-0.7, 0.7, size=2)
            ry, rx = rng.uniform(0.15, 0.8, size=2)
            d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
            with np.errstate(over="ignore"):
                m = 1.0 / (1.0 + np.exp((d - 1.0) * rng.uniform(4, 40)))
            col = rng.uniform(0, 1, size=3)
            img = img * (1 - m[..., None]) + col * m[..., None]
        img += rng.normal(0, rng.uniform(0.0, 0.06), size=img.shape)   # sensor noise
        out[i] = np.clip(np.rint(img * 255), 0, 255).astype(np.uint8)
    return out


def video_frame(h: int = 720, w: int = 1280, seed: int = 7) -> np.ndarray:
    """A seeded BGR 'video frame' (smooth gradients + noise), uint8 [h,w,3]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], axis=-1)
    return (base.astype(np.int32) + rng.integers(-20, 21, base.shape)).clip(0, 255).astype(np.uint8)


def head_boxes(k: int, h: int = 720, w: int = 1280, seed: int = 11) -> np.ndarray:
    """k seeded YOLO-style boxes (y_min, x_min, y_max, x_max), float32, inside an h x w frame."""
    rng = np.random.default_rng(seed)
    size = rng.uniform(0.06, 0.3, k) * h
    cy = rng.uniform(0.0, 1.0, k) * (h - size)
    cx = rng.uniform(0.0, 1.0, k) * (w - size * 0.8)
    return np.stack([cy, cx, cy + size, cx + size * 0.8], axis=1).astype(np.float32)


# yolo_v3/data/yolo_anchors.txt of the reference (w,h pairs, 9 anchors -> three output maps)
YOLO_ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326], np.float32).reshape(-1, 2)


def yolo_maps(seed: int, num_classes: int = 1, grid: int = 13, objects: int = 10, num_layers: int = 3):
    """Seeded stand-ins for the detector's output maps ([g,g,3*(5+C)], [2g,2g,..], [4g,4g,..], float32, coarsest
    first): low-confidence background plus `objects` planted detections, each repeated on neighbouring cells /
    anchors with slightly different offsets so that NMS has overlapping candidates to suppress."""
    rng = np.random.default_rng(seed)
    maps = []
    for l in range(num_layers):
        g = grid << l
        m = rng.normal(0.0, 1.0, (g, g, 3, 5 + num_classes)).astype(np.float32)
        m[..., 4] = rng.normal(-7.0, 1.5, (g, g, 3))                   # background confidence logits
        for _ in range(objects):
            y, x, a = int(rng.integers(1, g - 1)), int(rng.integers(1, g - 1)), int(rng.integers(0, 3))
            c = int(rng.integers(0, num_classes))
            for dy, dx, da in ((0, 0, 0), (0, 1, 0), (1, 0, 0), (0, 0, 1), (-1, 0, 2)):
                if rng.random() < 0.7 or (dy, dx, da) == (0, 0, 0):
                    yy, xx, aa = y + dy, x + dx, (a + da) % 3
                    t = m[yy, xx, aa]
                    t[0:2] = rng.normal(0.0, 0.6, 2) - 2.0 * np.array([dx, dy])     # pulls the centre back towards (x, y)
                    t[2:4] = rng.normal(0.0, 0.25, 2)
                    t[4] = rng.normal(3.0, 1.5)
                    t[5:] = rng.normal(-3.0, 1.0, num_classes)
                    t[5 + c] = rng.normal(3.0, 1.0)
        maps.append(m.reshape(g, g, 3 * (5 + num_classes)))
    return maps
