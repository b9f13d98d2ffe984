#!/usr/bin/env python3
"""Writes tests/golden/hf_features.npz: the 7x7x1280 backbone features of golden crops 0 and 1 as
computed by HuggingFace transformers' EfficientNet (float64) loaded with the synthetic snapshot --
an implementation that shares no code with oracle/ or the product.  Committed so that the GPU box
(where nothing under tests/ may need the build box's packages beyond numpy/torch) pins both the
oracle and the HIP path to it.  Run on the build box:  python tests/golden/make_hf_fixture.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
for p in (os.path.join(ROOT, "headposeestimation-whenet_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import whenet_oracle as O  # noqa: E402
from tests.hf_reference import hf_backbone_features  # noqa: E402
from whenet_hip import weights as W  # noqa: E402

idx = np.array([0, 1])
crops = np.load(os.path.join(HERE, "golden_crops.npy"))[idx]
w = W.synthetic(1234)
feat = hf_backbone_features(w, O.normalise(crops).astype(np.float64))
np.savez_compressed(os.path.join(HERE, "hf_features.npz"), crop_index=idx, features=feat.astype(np.float32))
print("hf_features.npz", feat.shape, float(np.abs(feat).max()))
