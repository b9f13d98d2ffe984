"""stem_fuse 1 against 0: logits of the golden crops (diagnostic)."""
import os, sys
import numpy as np
R = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(R, "headposeestimation-whenet_amd")); sys.path.insert(0, R)
import torch  # noqa
from whenet_hip import _lib, weights as W, synth
crops = np.load(os.path.join(R, "tests/golden/golden_crops.npy"))[:4]
with _lib.Handle(W.pack(W.synthetic(1234)), device=0, dtype=_lib.F16) as h:
    for fold in (1, 0):
        h.set_option("fold12", fold)
        h.set_option("stem_fuse", 1); y1, a1, l1 = h.forward(crops)
        h.set_option("stem_fuse", 0); y0, a0, l0 = h.forward(crops)
        print("fold12", fold, "equal", np.array_equal(l0, l1), "max |dlogit|", np.abs(l0 - l1).max(), "angles", np.abs(y0 - y1).max())
    # constant images: interior pixels identical -> edge handling shows up
    for val in (0, 128, 255):
        c = np.full((2, 224, 224, 3), val, np.uint8)
        h.set_option("fold12", 0)
        h.set_option("stem_fuse", 1); _, _, l1 = h.forward(c)
        h.set_option("stem_fuse", 0); _, _, l0 = h.forward(c)
        print("const", val, np.array_equal(l0, l1), np.abs(l0 - l1).max())
    rng = np.random.default_rng(0)
    tests = {}
    yy = (np.arange(224) % 256).astype(np.uint8)
    tests["rows"] = np.broadcast_to(yy[None, :, None, None], (2, 224, 224, 3)).copy()
    tests["cols"] = np.broadcast_to(yy[None, None, :, None], (2, 224, 224, 3)).copy()
    tests["chan"] = np.broadcast_to(np.array([10, 120, 250], np.uint8)[None, None, None, :], (2, 224, 224, 3)).copy()
    tests["noise"] = rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8)
    one = np.zeros((2, 224, 224, 3), np.uint8); one[:, 100, 57, 1] = 255
    tests["impulse"] = one
    for name, c in tests.items():
        h.set_option("stem_fuse", 1); _, _, l1 = h.forward(c)
        h.set_option("stem_fuse", 0); _, _, l0 = h.forward(c)
        print(name, np.array_equal(l0, l1), np.abs(l0 - l1).max())
