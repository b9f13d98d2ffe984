#!/usr/bin/env python3
"""f16 configuration vs the float64 oracle on seeded crops (the 48 crops of test_f16_accuracy_contract): max / mean /
p95 angle error and argmax flips, for every setting of the engine option front_impl (0 = front.hip on all blocks,
round 2's schedule; 1 = default per-layer choice; 2 = front2.hip on all blocks), and the f32 configuration."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
import numpy as np
from whenet_hip import _lib, synth, weights as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
w = W.synthetic(1234)
crops = np.concatenate([synth.scene_crops(n // 2, seed=5), synth.noise_crops(n - n // 2, seed=6)])
fx = os.path.join(ROOT, "tests", "golden", "f16_set_expected.npz")
if n == 48 and os.path.exists(fx):
    f = np.load(fx)
    ra, rl, rm = f["angles"], f["logits"], f["argmax"]
else:
    from oracle import whenet_oracle as O
    ref = O.forward(crops, w, np.float64)
    ra, rl, rm = np.stack([ref["yaw"], ref["pitch"], ref["roll"]], 1), ref["logits"], ref["argmax"]
for name, dt, impl in (("f16 front_impl=0", _lib.F16, 0), ("f16 front_impl=1 (default)", _lib.F16, 1), ("f16 front_impl=2", _lib.F16, 2),
                       ("f32", _lib.F32, 1)):
    h = _lib.Handle(W.pack(w), device=0, dtype=dt)
    h.set_option("front_impl", impl)
    y, a, l = h.forward(crops)
    e = np.abs(y - ra)
    print(f"{name:28s}: max {e.max():.4f} mean {e.mean():.5f} p95 {np.percentile(e, 95):.4f} deg; argmax flips {(a != rm).sum()} of {a.size}; "
          f"max |logit err| {np.abs(l - rl).max():.4f}")
    h.close()

# Larger sample, against the f32 configuration of the same library (itself within 1e-3 deg of the oracle): the f16
# schedule with and without option fold12 (block 1's project folded into block 2's expand weights).
big = np.concatenate([synth.scene_crops(256, seed=41), synth.noise_crops(256, seed=42)])
h32 = _lib.Handle(W.pack(w), device=0, dtype=_lib.F32)
y32, a32, l32 = h32.forward(big)
h32.close()
for fold in (0, 1):
    h = _lib.Handle(W.pack(w), device=0, dtype=_lib.F16)
    h.set_option("fold12", fold)
    y, a, l = h.forward(big)
    e = np.abs(y - y32)
    print(f"f16 fold12={fold} vs f32, 512 crops : max {e.max():.4f} mean {e.mean():.5f} p95 {np.percentile(e, 95):.4f} p99 {np.percentile(e, 99):.4f} deg; "
          f"argmax flips {(a != a32).sum()} of {a.size}; max |logit diff| {np.abs(l - l32).max():.4f} rms {np.sqrt(((l - l32) ** 2).mean()):.5f}")
    h.close()
