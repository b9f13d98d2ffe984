"""Second, independent CPU restatement of the WHENet hot path: torch float32, NCHW,
library convolutions (oneDNN).  TEST INFRASTRUCTURE -- parity unpinned (see the header of
oracle/whenet_oracle.py for why and for the reference file:line each stage follows).

Two jobs:
  1. an implementation that shares no code with oracle/whenet_oracle.py (different layout,
     different conv algorithm, float32) -- the two must agree to float32 noise
     (tests/test_oracle.py), which is how the restatement is checked in the absence of the
     reference's own runtime;
  2. the CPU baseline bench.py times beside the HIP path ("port": the true Keras-CPU path
     of /root/reference/whenet.py:22-34 cannot run here).  It includes the reference's float64
     normalisation (whenet.py:23-26), its batch_size=8 chunking (whenet.py:27) and its numpy
     softmax-expectation (whenet.py:28-33, utils.py:7-11).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import b0_spec as G       # the oracle's own geometry (nothing is imported from the product package)


def _same(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    _, pb, pa = G.tf_same(x.shape[2], k, s)
    _, qb, qa = G.tf_same(x.shape[3], k, s)
    if pb or pa or qb or qa:
        x = F.pad(x, (qb, qa, pb, pa))
    return x


class TorchWHENet:
    """Same surface as whenet.WHENet (/root/reference/whenet.py:6-34)."""

    def __init__(self, weights: Dict[str, np.ndarray], dtype: torch.dtype = torch.float32):
        self.dtype = dtype
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)  # noqa: E731
        self.p: Dict[str, torch.Tensor] = {}
        for name, a in weights.items():
            if name.endswith("/kernel") and a.ndim == 4:
                if name.endswith("/dw/kernel"):
                    self.p[name] = t(np.transpose(a, (2, 3, 0, 1)))      # (C,1,kh,kw)
                else:
                    self.p[name] = t(np.transpose(a, (3, 2, 0, 1)))      # (O,I,kh,kw)
            else:
                self.p[name] = t(a)
        self.idx_tensor = np.arange(66, dtype=np.float32)                # whenet.py:17-18
        self.idx_tensor_yaw = np.arange(120, dtype=np.float32)           # whenet.py:19-20

    def _bn(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        p = self.p
        return F.batch_norm(x, p[f"{prefix}/mean"], p[f"{prefix}/var"], p[f"{prefix}/gamma"],
                            p[f"{prefix}/beta"], training=False, eps=G.BN_EPSILON)

    @torch.no_grad()
    def logits(self, x_nhwc_f32: np.ndarray) -> List[np.ndarray]:
        """Model.predict body: normalised float32 NHWC -> [yaw, pitch, roll] logits."""
        p = self.p
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc_f32)).to(self.dtype).permute(0, 3, 1, 2).contiguous()
        x = F.silu(self._bn(F.conv2d(_same(x, 3, 2), p["stem/conv/kernel"], stride=2), "stem/bn"))
        for b in G.mbconv_blocks():
            q = f"b{b.number}"
            inp = x
            if b.expands:
                x = F.silu(self._bn(F.conv2d(x, p[f"{q}/expand/kernel"]), f"{q}/expand_bn"))
            x = F.conv2d(_same(x, b.kernel, b.stride), p[f"{q}/dw/kernel"], stride=b.stride, groups=b.filters_mid)
            x = F.silu(self._bn(x, f"{q}/dw_bn"))
            sq = x.mean(dim=(2, 3), keepdim=True)
            r = F.silu(F.conv2d(sq, p[f"{q}/se_reduce/kernel"], p[f"{q}/se_reduce/bias"]))
            g = torch.sigmoid(F.conv2d(r, p[f"{q}/se_expand/kernel"], p[f"{q}/se_expand/bias"]))
            x = x * g
            x = self._bn(F.conv2d(x, p[f"{q}/project/kernel"]), f"{q}/project_bn")
            if b.identity_skip:
                x = x + inp
        x = F.silu(self._bn(F.conv2d(x, p["head/conv/kernel"]), "head/bn"))
        f = x.mean(dim=(2, 3))
        return [(f @ p[f"{n}/kernel"] + p[f"{n}/bias"]).to(torch.float32).numpy() for n in ("yaw", "pitch", "roll")]

    def predict_logits(self, img: np.ndarray, batch_size: int = 8) -> List[np.ndarray]:
        """keras Model.predict(img, batch_size=8) (whenet.py:27): float32 cast + chunking."""
        x = np.asarray(img, dtype=np.float32)
        outs: List[List[np.ndarray]] = [[], [], []]
        for i in range(0, x.shape[0], batch_size):
            for k, o in enumerate(self.logits(x[i:i + batch_size])):
                outs[k].append(o)
        return [np.concatenate(o, axis=0) for o in outs]

    def get_angle(self, img, batch_size: int = 8) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """whenet.py:22-34, statement for statement (numpy pre/post, float64 normalise)."""
        mean = [0.485, 0.456, 0.406]
        std = [0.229, 0.224, 0.225]
        img = img / 255
        img = (img - mean) / std
        predictions = self.predict_logits(img, batch_size=batch_size)
        yaw_predicted = _softmax(predictions[0])
        pitch_predicted = _softmax(predictions[1])
        roll_predicted = _softmax(predictions[2])
        yaw_predicted = np.sum(yaw_predicted * self.idx_tensor_yaw, axis=1) * 3 - 180
        pitch_predicted = np.sum(pitch_predicted * self.idx_tensor, axis=1) * 3 - 99
        roll_predicted = np.sum(roll_predicted * self.idx_tensor, axis=1) * 3 - 99
        return yaw_predicted, pitch_predicted, roll_predicted

    predict = get_angle


def _softmax(x: np.ndarray) -> np.ndarray:
    """utils.py:7-11."""
    x -= np.max(x, axis=1, keepdims=True)
    a = np.exp(x)
    b = np.sum(np.exp(x), axis=1, keepdims=True)
    return a / b
