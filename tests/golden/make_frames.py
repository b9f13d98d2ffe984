"""Freeze the reference's two sample images as decoded frames + the crops its demo.py would cut.

Run HERE (the container that has /root/reference); the GPU box only reads the .npz.
  python tests/golden/make_frames.py

demo.py:7-11: cv2.imread (BGR) -> cvtColor(BGR2RGB) -> img[y_min:y_max, x_min:x_max] with the
integer bbox of Sample/bbox.txt (x_min y_min x_max y_max) -> cv2.resize(.., (224, 224)).
cv2 is not installable here: the JPEGs are decoded with PIL (libjpeg, as cv2.imread uses) and the
resize is the oracle's restatement of OpenCV's fixed-point INTER_LINEAR (parity unpinned, see
oracle/preprocess_oracle.py).
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import preprocess_oracle as P   # noqa: E402

REF = "/root/reference/Sample"
out = {}
with open(os.path.join(REF, "bbox.txt")) as f:
    lines = [l.strip() for l in f if l.strip()]
for i, l in enumerate(lines):
    name, bbox = l.split(",")
    x_min, y_min, x_max, y_max = (int(b) for b in bbox.split(" "))
    rgb = np.array(Image.open(os.path.join(REF, name)).convert("RGB"), np.uint8)
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])                 # what cv2.imread returns
    rect = np.array([y_min, x_min, y_max, x_max], np.int32)      # demo.py:10 window
    out[f"frame{i}"] = bgr
    out[f"rect{i}"] = rect
    out[f"crop{i}"] = P.crop_and_resize(bgr, rect, bgr2rgb=True)
np.savez_compressed(os.path.join(HERE, "sample_frames.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
