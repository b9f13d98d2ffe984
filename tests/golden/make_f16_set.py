#!/usr/bin/env python3
"""Expected outputs (float64 oracle, oracle/whenet_oracle.py) for the 48 seeded crops of the f16 accuracy
contract (tests/test_gpu_parity.py::test_f16_accuracy_contract): 24 scene crops (seed 5) + 24 noise crops (seed 6),
the seeded synthetic snapshot 1234.  Writes tests/golden/f16_set_expected.npz (angles [48,3], logits [48,252],
argmax [48,3]).  Usage: python tests/golden/make_f16_set.py"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
sys.path.insert(0, ROOT)
from oracle import whenet_oracle as O          # noqa: E402
from whenet_hip import synth, weights as W     # noqa: E402

crops = np.concatenate([synth.scene_crops(24, seed=5), synth.noise_crops(24, seed=6)])
ref = O.forward(crops, W.synthetic(1234), np.float64)
out = os.path.join(ROOT, "tests", "golden", "f16_set_expected.npz")
np.savez_compressed(out, angles=np.stack([ref["yaw"], ref["pitch"], ref["roll"]], 1).astype(np.float64),
                    logits=ref["logits"].astype(np.float32), argmax=ref["argmax"].astype(np.int32))
print("wrote", out, os.path.getsize(out), "bytes")
