#!/usr/bin/env python3
"""Where does the f16 configuration's angle error come from?    TEST INFRASTRUCTURE.

A float64 re-run of the oracle's backbone (oracle/whenet_oracle.py primitives) with a switchable
round-to-binary16 at exactly the places where the HIP f16 path rounds:
  W     BN-folded 1x1-conv weights (the MFMA operands; depthwise taps, biases, SE and the Dense heads stay f32)
  stem  stem output          E   expanded tensor      D   depthwise output
  Gg    SE gate (stored in T since round 2)        Gp  the gated product D*g fed to the project MFMA
  X     block outputs (the residual stream)            H   head-conv output
and prints max / mean |angle - f64| over the crops for a few combinations -- among them the
"f32 residual trunk" (everything but X rounded).  Usage: python tests/f16_error_study.py [ncrops]
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
from oracle import b0_spec as G                      # noqa: E402
from oracle import whenet_oracle as O                # noqa: E402
from whenet_hip import synth, weights as W          # noqa: E402


def r16(t):
    return t.astype(np.float16).astype(np.float64)


def folded(w, conv, bn, rnd):
    k = w[conv].astype(np.float64)
    s = w[f"{bn}/gamma"].astype(np.float64) / np.sqrt(w[f"{bn}/var"].astype(np.float64) + G.BN_EPSILON)
    b = w[f"{bn}/beta"].astype(np.float64) - w[f"{bn}/mean"].astype(np.float64) * s
    k = k * s                                         # output channel is the last axis for HWIO and HWC1 alike
    return (r16(k) if rnd else k), b


def backbone(x, w, on):
    q = lambda tag, t: r16(t) if tag in on else t     # noqa: E731
    k, b = folded(w, "stem/conv/kernel", "stem/bn", False)
    x = q("stem", O.swish(O.conv2d(x, k, 2) + b))
    for blk in G.mbconv_blocks():
        p = f"b{blk.number}"
        inp = x
        if blk.expands:
            k, b = folded(w, f"{p}/expand/kernel", f"{p}/expand_bn", "W" in on)
            x = q("E", O.swish(O.conv2d(x, k, 1) + b))
        s = w[f"{p}/dw_bn/gamma"].astype(np.float64) / np.sqrt(w[f"{p}/dw_bn/var"].astype(np.float64) + G.BN_EPSILON)
        kd = w[f"{p}/dw/kernel"].astype(np.float64) * s[None, None, :, None]
        bd = w[f"{p}/dw_bn/beta"].astype(np.float64) - w[f"{p}/dw_bn/mean"].astype(np.float64) * s
        x = O.swish(O.depthwise(x, kd, blk.stride) + bd)
        sq = x.mean(axis=(1, 2), keepdims=True)       # (the kernels sum the f32 values before rounding D)
        x = q("D", x)
        r = O.swish(sq @ w[f"{p}/se_reduce/kernel"][0, 0].astype(np.float64) + w[f"{p}/se_reduce/bias"])
        g = q("Gg", O.sigmoid(r @ w[f"{p}/se_expand/kernel"][0, 0].astype(np.float64) + w[f"{p}/se_expand/bias"]))
        x = q("Gp", x * g)
        k, b = folded(w, f"{p}/project/kernel", f"{p}/project_bn", "W" in on)
        x = O.conv2d(x, k, 1) + b
        if blk.identity_skip:
            x = x + inp
        x = q("X", x)
    k, b = folded(w, "head/conv/kernel", "head/bn", "W" in on)
    return q("H", O.swish(O.conv2d(x, k, 1) + b))


def angles(u8, w, on):
    out = []
    for i in range(0, u8.shape[0], 8):
        f = backbone(O.normalise(u8[i:i + 8]).astype(np.float64), w, on)
        out.append(np.stack(O.decode(O.heads(f, w)), axis=1))
    return np.concatenate(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    w = W.synthetic(1234)
    u8 = np.concatenate([synth.scene_crops(n // 2, seed=5), synth.noise_crops(n - n // 2, seed=6)])
    ref = angles(u8, w, set())
    ALL = {"W", "stem", "E", "D", "Gg", "Gp", "X", "H"}
    rows = [("everything the HIP f16 path rounds", ALL),
            ("f32 residual trunk (X kept f32)", ALL - {"X"}),
            ("f32 residual trunk + f32 gate and product", ALL - {"X", "Gg", "Gp"}),
            ("gate kept f32, product rounded (round 1)", ALL - {"Gg"}),
            ("only the residual stream X", {"X"}),
            ("only weights W", {"W"}),
            ("only E", {"E"}), ("only D", {"D"}), ("only gate + gated product", {"Gg", "Gp"}), ("only the gate", {"Gg"}),
            ("only stem + head", {"stem", "H"}),
            ("all activations, f32 weights", ALL - {"W"})]
    print(f"{n} crops, synthetic weights seed 1234; |angle - f64 oracle| in degrees")
    for name, on in rows:
        e = np.abs(angles(u8, w, on) - ref)
        print(f"  {name:42s} max {e.max():7.4f}   mean {e.mean():7.4f}   p95 {np.percentile(e, 95):7.4f}")


if __name__ == "__main__":
    main()
