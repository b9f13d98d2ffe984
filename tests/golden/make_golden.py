#!/usr/bin/env python3
"""Regenerates every fixture under tests/golden/.  Run in the BUILD container only
(it reads /root/reference/Sample/*; nothing at test / bench time does):

    python tests/golden/make_golden.py

Outputs
  sample_crops.npy      [2,224,224,3] u8: the reference's two sample inputs
                        (/root/reference/Sample/bbox.txt:1-2), cropped as demo.py:9-11 does --
                        but decoded and resized with PIL (bilinear), because cv2 is not
                        installable here; parity is defined on identical uint8 crops, so
                        both the oracle and the HIP path consume these bytes.
  ../../headposeestimation-whenet_amd/whenet_hip/data/calib_seed1234.npz
                        BN moving statistics + head calibration of the seeded synthetic
                        snapshot (whenet_hip/weights.py::synthetic; shipped inside the package, the
                        product never reads tests/) -- the trained WHENet.h5 is absent from the
                        reference (.MISSING_LARGE_BLOBS:1).
  golden_crops.npy      [8,224,224,3] u8: 2 sample + 4 scene + 2 noise crops.
  golden_expected.npz   float64-oracle logits / angles / argmax / top-2 margins for those
                        crops, plus the float32 restatements' deviations (noise floor).
  golden.json           sha256 of the snapshot + summary numbers.

There are no reference-side golden vectors to pin against (the reference has no tests and
its runtime is not installable): these are *self-generated* known answers -- "parity
unpinned" (oracle/whenet_oracle.py header).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
sys.path.insert(0, ROOT)

from whenet_hip import spec, synth, weights  # noqa: E402
from oracle import whenet_oracle as O  # noqa: E402

SEED = 1234
REF_SAMPLE = "/root/reference/Sample"


def sample_crops() -> np.ndarray:
    from PIL import Image
    crops = []
    with open(os.path.join(REF_SAMPLE, "bbox.txt")) as f:
        for line in f.read().splitlines():
            if not line.strip():
                continue
            name, bbox = line.split(",")
            x0, y0, x1, y1 = [int(v) for v in bbox.split(" ")]
            im = Image.open(os.path.join(REF_SAMPLE, name)).convert("RGB")
            arr = np.asarray(im)[y0:y1, x0:x1]                     # demo.py:9-10
            rs = Image.fromarray(arr).resize((224, 224), Image.BILINEAR)   # demo.py:11 (cv2 INTER_LINEAR)
            crops.append(np.asarray(rs, dtype=np.uint8))
    return np.stack(crops)


def calibrate(w, crops):
    """Forward in float64; at every BN install the statistics of its own input."""
    calib = {}

    def hook(prefix, t):
        m = t.mean(axis=(0, 1, 2))
        v = t.var(axis=(0, 1, 2))
        v = np.maximum(v, 1e-4)
        w[f"{prefix}/mean"] = m.astype(np.float32)
        w[f"{prefix}/var"] = v.astype(np.float32)
        calib[f"{prefix}/mean"] = w[f"{prefix}/mean"]
        calib[f"{prefix}/var"] = w[f"{prefix}/var"]

    x = O.normalise(crops).astype(np.float64)
    # statistics are over the whole calibration set: one forward, all crops at once.
    f = O.backbone(x, w, bn_hook=hook)
    feat = f.mean(axis=(1, 2))
    calib["feat_mean"] = feat.mean(axis=0).astype(np.float32)
    # heads: mu = c + gain * u.(f - fmean) should stay within +-n/10 bins of the centre for
    # every calibration crop (iid-noise crops are the outliers that set this);
    # the smooth perturbation should move logits by ~1.0 (std over crops and bins).
    dev = feat - calib["feat_mean"].astype(np.float64)
    basis = weights.head_basis(SEED)
    for name, n in (("yaw", spec.N_YAW), ("pitch", spec.N_PITCH), ("roll", spec.N_ROLL)):
        t = dev @ basis[name]["u"]
        calib[f"{name}/gain"] = np.float32((n / 10.0) / np.abs(t).max())
        calib[f"{name}/pert"] = np.float32(1.0 / (dev @ basis[name]["smooth"]).std())
    return calib


def main():
    os.makedirs(HERE, exist_ok=True)
    sc = sample_crops()
    np.save(os.path.join(HERE, "sample_crops.npy"), sc)

    cal_crops = np.concatenate([sc, synth.scene_crops(8, seed=100), synth.noise_crops(4, seed=101)])
    w = weights.synthetic_raw(SEED)
    calib = calibrate(w, cal_crops)
    np.savez(os.path.join(ROOT, "headposeestimation-whenet_amd", "whenet_hip", "data", f"calib_seed{SEED}.npz"), **calib)

    w = weights.synthetic(SEED)
    bb, hd = spec.param_count()
    assert (bb, hd) == (4_049_564, 322_812)

    gold = np.concatenate([sc, synth.scene_crops(4, seed=7), synth.noise_crops(2, seed=0)])
    np.save(os.path.join(HERE, "golden_crops.npy"), gold)
    r64 = O.forward(gold, w, np.float64)
    r32 = O.forward(gold, w, np.float32)
    from oracle.whenet_torch import TorchWHENet
    tw = TorchWHENet(w)
    ty, tp, tr = tw.get_angle(gold.copy())
    tl = np.concatenate(tw.predict_logits(O.normalise(gold)), axis=1)
    ang64 = np.stack([r64["yaw"], r64["pitch"], r64["roll"]], axis=1)
    ang32 = np.stack([r32["yaw"], r32["pitch"], r32["roll"]], axis=1)
    angt = np.stack([ty, tp, tr], axis=1)
    margins = O.top2_margin(r64["logits"])
    np.savez(os.path.join(HERE, "golden_expected.npz"),
             logits=r64["logits"], angles=ang64, argmax=r64["argmax"], margins=margins,
             logits_np32=r32["logits"].astype(np.float32), angles_np32=ang32.astype(np.float32),
             logits_torch32=tl.astype(np.float32), angles_torch32=angt.astype(np.float32))
    info = {
        "seed": SEED,
        "weights_sha256": weights.checksum(w),
        "params": {"backbone": bb, "heads": hd},
        "angles_deg_f64": ang64.tolist(),
        "argmax": r64["argmax"].tolist(),
        "min_top2_margin": float(margins.min()),
        "noise_floor_deg": {
            "numpy_f32_vs_f64": float(np.abs(ang32 - ang64).max()),
            "torch_f32_vs_f64": float(np.abs(angt - ang64).max()),
        },
        "noise_floor_logits": {
            "numpy_f32_vs_f64": float(np.abs(r32["logits"] - r64["logits"]).max()),
            "torch_f32_vs_f64": float(np.abs(tl - r64["logits"]).max()),
        },
        "logit_std": float(r64["logits"].std()),
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(json.dumps(info, indent=1))


if __name__ == "__main__":
    main()
