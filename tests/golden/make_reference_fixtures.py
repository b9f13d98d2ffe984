#!/usr/bin/env python3
"""Executes the REFERENCE'S OWN CODE (tests/refharness.py) and records what it returns.

    python tests/golden/make_reference_fixtures.py          # BUILD container only (reads /root/reference)

The GPU box has no /root/reference, so what the reference computes is committed here as fixtures and
the -m gpu tests compare the HIP path with THESE arrays:

  reference_get_angle.npz   /root/reference/whenet.py:7-34 + utils.py:7-11 executed as written on the
                            golden crops for N = 0, 1, 7, 8, 9, 64 and on real-valued (non-byte) input:
                            yaw/pitch/roll exactly as `WHENet.get_angle` returned them (float32), the three
                            arrays `Model.predict` handed back (argmax is taken from those), and the
                            (N, batch_size) of the predict call.  Only the body of efn.EfficientNetB0 --
                            third-party, not in the reference -- is the float64 restatement
                            (oracle/whenet_oracle.backbone): the BACKBONE STAYS UNPINNED.
  reference_rects.npz       /root/reference/demo_video.py:11-35 (`process_detection`) executed as written
                            on >= 1,000 float32 YOLO boxes over four frame sizes, with a frame object that
                            records the slice `img[int(y_min):int(y_max), int(x_min):int(x_max)]` it is asked
                            for: the window arithmetic incl. its order dependence and clipping.
  reference_yolo.npz        /root/reference/yolo_v3/model.py:125-232 (`yolo_head`, `yolo_correct_boxes`,
                            `yolo_boxes_and_scores`, `yolo_eval`) executed as written over a numpy float32
                            `keras.backend`; only tf.image.non_max_suppression is the restatement.
                            Every decoded box/score (what yolo_eval masks) and the final selection.
  reference_demo.json       /root/reference/demo.py:19-30 executed as written against the drop-in module
                            `whenet` (CPU: the handle behind it is the float64 oracle): the call shapes the
                            demo uses and the angles it drew for the two Sample/ images.
"""
import hashlib
import json
import os
import runpy
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
for p in (os.path.join(ROOT, "headposeestimation-whenet_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from whenet_hip import synth, weights  # noqa: E402
from oracle import whenet_oracle as O  # noqa: E402
from oracle import yolo_oracle as Y  # noqa: E402
from oracle import preprocess_oracle as P  # noqa: E402
from tests import refharness as H  # noqa: E402

from tests.refcases import SIZES, FRAMES, YOLO_CASES, crops64, real_valued, boxes_for, all_bytes_image, lut_from_normalised, yolo_case_maps  # noqa: E402

SEED = 1234


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------------ get_angle
def run_get_angle(w):
    all_crops = crops64()
    out = {"crops_sha256": np.array(sha(all_crops)), "sizes": np.array(SIZES)}
    for n in SIZES:
        y, p, r, m = H.run_get_angle(all_crops[:n], w, O.backbone)
        assert m.model.predict_calls == [(n, 8)]                                     # whenet.py:27
        assert [o.layer.name for o in m.model.outputs] == ["yaw_new", "pitch_new", "roll_new"]   # whenet.py:11-14
        assert np.array_equal(m.idx_tensor, np.arange(66, dtype=np.float32))
        assert np.array_equal(m.idx_tensor_yaw, np.arange(120, dtype=np.float32))
        out[f"n{n}_angles"] = np.stack([y, p, r], axis=1)
        out[f"n{n}_dtype"] = np.array(str(y.dtype))
        out[f"n{n}_logits"] = np.concatenate(m.model.last_outputs, axis=1)
        print(f"get_angle N={n}: dtype {y.dtype}, first {out[f'n{n}_angles'][:1].tolist()}", flush=True)
    x = real_valued(all_crops)
    y, p, r, m = H.run_get_angle(x, w, O.backbone)
    out["real_angles"] = np.stack([y, p, r], axis=1)
    out["real_logits"] = np.concatenate(m.model.last_outputs, axis=1)
    # whenet.py:23-26 on every byte value: what Model.predict is handed (after Keras' float32 cast)
    seen = []

    def spy(xn, ww):
        seen.append(xn.astype(np.float32))
        return np.zeros((xn.shape[0], 7, 7, 1280))

    H.run_get_angle(all_bytes_image(), w, spy, compute_dtype=np.float32)
    out["normalise_lut"] = lut_from_normalised(seen[0])
    np.savez_compressed(os.path.join(HERE, "reference_get_angle.npz"), **out)
    return out


# ------------------------------------------------------------------------------------------------ rects
def run_rects():
    class Args:
        display = "full"                                                                # also runs demo_video.py:31-34

    class FakeModel:
        def get_angle(self, img):
            assert img.shape == (1, 224, 224, 3)                                        # demo_video.py:24
            return np.array([10.0], np.float32), np.array([-5.0], np.float32), np.array([2.5], np.float32)

    boxes, hw, raw, eff, rect_args = [], [], [], [], []
    with H.reference() as R:
        dv = R.load("demo_video")
        for (h, w) in FRAMES:
            for b in boxes_for(h, w):
                frame = H.RecordingFrame(h, w)
                del R.cv2.calls[:]
                dv.process_detection(FakeModel(), frame, b, Args)
                (sy, sx), = frame.slices
                assert sy.step is None and sx.step is None
                raw.append([sy.start, sx.start, sy.stop, sx.stop])
                y0, y1, _ = sy.indices(h)
                x0, x1, _ = sx.indices(w)
                eff.append([y0, x0, max(y1, y0), max(x1, x0)])
                rect = [c for c in R.cv2.calls if c[0] == "rectangle"][0]
                rect_args.append([rect[1][0], rect[1][1], rect[2][0], rect[2][1]])       # (x_min, y_min), (x_max, y_max)
                boxes.append(b)
                hw.append([h, w])
    out = dict(boxes=np.array(boxes, np.float32), frame_hw=np.array(hw, np.int32), slice_raw=np.array(raw, np.int64),
               rects=np.array(eff, np.int32), rectangle=np.array(rect_args, np.int64))
    np.savez_compressed(os.path.join(HERE, "reference_rects.npz"), **out)
    print(f"rects: {len(boxes)} boxes; clipped at a frame edge: "
          f"{int(((out['rects'][:, :2] == 0).any(1) | (out['rects'][:, 2:] == out['frame_hw']).any(1)).sum())}", flush=True)
    return out


# ------------------------------------------------------------------------------------------------ yolo
def run_yolo():
    out = {}
    with H.reference(nms_fn=Y.non_max_suppression) as R:
        ym = R.load("yolo_v3.model")
        with open(os.path.join(R.dir, "yolo_v3", "data", "yolo_anchors.txt")) as f:     # yolo_postprocess.py:62-66
            anchors = np.array([float(x) for x in f.readline().split(",")]).reshape(-1, 2)
        out["anchors"] = anchors.astype(np.float32)
        for i, case in enumerate(YOLO_CASES):
            seed, nc, image, max_boxes, score, iou = case[:6]
            maps, case_anchors = yolo_case_maps(case)
            assert np.array_equal(case_anchors, anchors[:len(case_anchors)].astype(np.float32))
            del R.tf.masked[:]
            b, s, c = ym.yolo_eval([m[None] for m in maps], anchors[:len(case_anchors)], nc, np.array(image), max_boxes=max_boxes,
                                   score_threshold=score, iou_threshold=iou)
            all_boxes, mask0 = R.tf.masked[0]                       # model.py:219: boolean_mask(boxes, mask[:, 0])
            all_scores = np.stack([R.tf.masked[2 * k + 1][0] for k in range(nc)], axis=1)
            out[f"case{i}_cfg"] = np.array([seed, nc, image[0], image[1], max_boxes])
            out[f"case{i}_thr"] = np.array([score, iou], np.float64)
            out[f"case{i}_all_boxes"] = all_boxes.astype(np.float32)
            out[f"case{i}_all_scores"] = all_scores.astype(np.float32)
            out[f"case{i}_boxes"], out[f"case{i}_scores"], out[f"case{i}_classes"] = b, s, c
            assert b.dtype == np.float32 and s.dtype == np.float32, (b.dtype, s.dtype)
            print(f"yolo case {i}: {all_boxes.shape[0]} candidates -> {len(b)} detections, classes {sorted(set(c.tolist()))}",
                  flush=True)
    np.savez_compressed(os.path.join(HERE, "reference_yolo.npz"), **out)
    return out


# ------------------------------------------------------------------------------------------------ demo.py
class OracleHandle:
    """What stands behind the drop-in class when demo.py is run on CPU: same methods as whenet_hip._lib.Handle
    that whenet.WHENet touches, arithmetic by the float64 oracle."""
    calls = []

    def __init__(self, snapshot, device=0, dtype=0):
        self.w = weights.load(snapshot) if isinstance(snapshot, str) else weights.unpack(snapshot)

    def forward(self, u8, want_logits=False):
        r = O.forward(u8, self.w)
        ypr = np.stack([r["yaw"], r["pitch"], r["roll"]], axis=1).astype(np.float32)
        OracleHandle.calls.append((list(u8.shape), str(u8.dtype), ypr[0].tolist(), hashlib.sha256(u8.tobytes()).hexdigest()))
        return ypr, r["argmax"], r["logits"].astype(np.float32)

    def info(self):
        import types
        from whenet_hip import _lib
        return types.SimpleNamespace(params_backbone=4_049_564, params_heads=322_812, device_name=b"cpu oracle",
                                     arch=b"none", dtype=_lib.F32, n_kernels_per_forward=0)

    def close(self):
        pass


def run_demo(w, write=True):
    from whenet_hip import _lib
    tmp = tempfile.mkdtemp(prefix="whenet_demo_")
    weights.save(os.path.join(tmp, "WHENet.h5"), w)                    # demo.py:20 opens 'WHENet.h5' in the cwd
    os.symlink(os.path.join(H.REF, "Sample"), os.path.join(tmp, "Sample"))
    cwd = os.getcwd()
    real_handle = _lib.Handle
    OracleHandle.calls = []
    try:
        os.chdir(tmp)
        _lib.Handle = OracleHandle
        with H.reference(resize_fn=P.resize_linear_u8, dropin_whenet=True) as R:
            runpy.run_path(os.path.join(R.dir, "demo.py"), run_name="__main__")
            calls = list(R.cv2.calls)
    finally:
        _lib.Handle = real_handle
        os.chdir(cwd)
    lines = [c for c in calls if c[0] == "line"]
    rects = [c for c in calls if c[0] == "rectangle"]
    out = {"forward_calls": OracleHandle.calls, "n_lines": len(lines), "rectangles": [list(map(list, r[1:3])) for r in rects],
           "lines": [[list(c[1]), list(c[2]), list(c[3])] for c in lines], "waitKey": [c[1] for c in calls if c[0] == "waitKey"]}
    if write:
        with open(os.path.join(HERE, "reference_demo.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("demo.py:", json.dumps(out)[:400], flush=True)
    return out


def main():
    w = weights.synthetic(SEED)
    which = set(sys.argv[1:]) or {"rects", "yolo", "demo", "get_angle"}
    if "rects" in which:
        run_rects()
    if "yolo" in which:
        run_yolo()
    if "demo" in which:
        run_demo(w)
    if "get_angle" in which:
        run_get_angle(w)


if __name__ == "__main__":
    main()
