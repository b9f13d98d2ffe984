#!/usr/bin/env python3
"""Layer-by-layer diagnostic of the HIP path against the float64 oracle (run on the GPU box):

    python tools/gpu_diag.py [--dtypes f32,f16] [--quick]

For every dtype and both 1x1-conv implementations (MFMA / scalar check kernels) it prints the
relative error of every kernel output when fed the oracle's input for that layer, then the
end-to-end angle/logit errors on the golden crops.  Never raises: failures are printed so one
GPU call yields the whole picture.
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np
import torch  # noqa: F401  (first: single HIP runtime)

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
sys.path.insert(0, ROOT)

from oracle import whenet_oracle as O  # noqa: E402
from whenet_hip import _lib, spec, weights as W  # noqa: E402


def rel(got, ref):
    ref = np.asarray(ref, np.float64)
    sc = max(np.sqrt((ref ** 2).mean()), 1e-9)
    d = np.abs(np.asarray(got, np.float64) - ref)
    return d.max() / sc, np.sqrt((d ** 2).mean()) / sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="f32,f16")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(),
          torch.cuda.get_device_name(0) if torch.cuda.is_available() else "-")
    w = W.synthetic(1234)
    blob = W.pack(w)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_crops.npy"))
    exp = dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_expected.npz")))
    crops = gold[[0, 3]]
    taps = {}
    f = O.backbone(O.normalise(crops).astype(np.float64), w, taps=taps)
    logits_ref = O.heads(f, w)

    for name in args.dtypes.split(","):
        dt = _lib.F16 if name == "f16" else _lib.F32
        try:
            h = _lib.Handle(blob, device=0, dtype=dt)
        except Exception:
            traceback.print_exc()
            continue
        i = h.info()
        print(f"\n===== dtype {name}  device {i.device_name.decode()} {i.arch.decode()} CUs {i.compute_units}")
        for impl in (0, 1):
            try:
                h.set_option("pw_impl", impl)
                print(f"--- pw_impl={impl} ({'MFMA' if impl == 0 else 'scalar check'}) : max-rel / rms-rel error per kernel")
                e = rel(h.op_stem(crops), taps["stem"])
                print(f"stem            {e[0]:.2e} {e[1]:.2e}")
                for b in spec.blocks():
                    p = f"b{b.index}"
                    x = taps["stem"] if b.index == 1 else taps[f"b{b.index - 1}/out"]
                    r = h.op_block(b.index, x.astype(np.float32))
                    parts = []
                    if b.has_expand and impl == 1:
                        # (with the MFMA path the expand conv is fused into the front kernel: the
                        # expanded tensor only exists in LDS, there is nothing to compare)
                        parts.append("exp %.2e" % rel(r["expand"], taps[f"{p}/expand"])[0])
                    parts.append("dw %.2e" % rel(r["dw"], taps[f"{p}/dw"])[0])
                    parts.append("gate %.2e" % rel(r["gate"], taps[f"{p}/gate"].reshape(r["gate"].shape))[0])
                    parts.append("out %.2e" % rel(r["out"], taps[f"{p}/out"])[0])
                    print(f"{p:4s} k{b.k}s{b.s} {b.h_in:3d}->{b.h_out:3d} C{b.cin}->{b.cexp}->{b.cout}: " + "  ".join(parts))
                r = h.op_head(taps["b16/out"].astype(np.float32))
                print("head: feat %.2e  logits abs %.2e" % (rel(r["feat"], taps["head"].mean(axis=(1, 2)))[0],
                                                            np.abs(r["logits"] - logits_ref).max()))
                t0 = time.perf_counter()
                ypr, am, lg = h.forward(gold)
                dt_s = time.perf_counter() - t0
                print("e2e golden: max |angle err| %.3e deg  max |logit err| %.3e  argmax mismatches %d/%d  (%.1f ms)" % (
                    np.abs(ypr - exp["angles"]).max(), np.abs(lg - exp["logits"]).max(),
                    int((am != exp["argmax"]).sum()), am.size, dt_s * 1e3))
                print("   per-crop angle err:", np.array2string(np.abs(ypr - exp["angles"]).max(1), precision=5))
            except Exception:
                traceback.print_exc()
        try:
            h.set_option("pw_impl", 0)
            if not args.quick:
                for n in (1, 8, 64):
                    c = np.concatenate([gold] * ((n + 7) // 8))[:n]
                    h.forward(c)
                    t0 = time.perf_counter()
                    reps = 20
                    for _ in range(reps):
                        h.forward(c, want_logits=False)
                    el = (time.perf_counter() - t0) / reps
                    print(f"host-pointer forward n={n}: {el * 1e3:.3f} ms  ({n / el:.0f} crops/s incl. H2D/D2H)")
        except Exception:
            traceback.print_exc()
        h.close()
    print("diag done")


if __name__ == "__main__":
    main()
