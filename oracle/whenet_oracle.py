"""CPU oracle for the WHENet hot path -- numpy, float64 by default.  TEST INFRASTRUCTURE.

**Parity: pinned to EXECUTED reference code for everything the reference wrote itself; the backbone is unpinned.**
normalise(), softmax(), decode(), the head wiring and the batch_size=8 chunking are checked against what
/root/reference/whenet.py:7-34 and utils.py:7-11 return when they are RUN (tests/refharness.py puts stand-ins for
keras / efficientnet / cv2 into sys.modules; tests/golden/make_reference_fixtures.py commits the outputs as
tests/golden/reference_get_angle.npz; tests/test_reference_run.py asserts oracle == reference-run output on CPU and the
HIP path against the same arrays on the GPU).  What that run cannot cover is the body of ``efn.EfficientNetB0``: the
reference has no tests, no golden vectors and no recorded outputs, its arithmetic lives in un-vendored third-party
packages (``efficientnet==0.0.4`` on ``keras==2.1.6`` / ``tensorflow-gpu==1.12.0``,
/root/reference/requirements.txt:3-5) that are not installable here, and the trained snapshot ``WHENet.h5`` is absent
(/root/reference/.MISSING_LARGE_BLOBS:1).  backbone() is a restatement of the published algorithm anchored on the
reference's call site (whenet.py:8); it is cross-checked against two independent implementations
(oracle/whenet_torch.py, and the HuggingFace ``transformers`` EfficientNet in tests/test_oracle.py), not against the
reference itself: **backbone parity unpinned**.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (headposeestimation-whenet_amd/) never does.

What each function follows:
  normalise()        /root/reference/whenet.py:23-26  (float64 arithmetic, then Keras casts
                     the array to float32 inside Model.predict, whenet.py:27)
  backbone()         whenet.py:8 -> efficientnet 0.0.4 EfficientNetB0(include_top=False):
                     stem Conv3x3/s2 'same' + BN + Swish; 16 MBConvBlock (expand 1x1 + BN +
                     Swish | depthwise kxk 'same' + BN + Swish | SEBlock on
                     int(input_filters*0.25) | project 1x1 + BN | identity skip when
                     stride 1 and in==out; DropConnect is identity at inference);
                     head Conv1x1(1280) + BN + Swish.   
  heads()            whenet.py:9-13  GlobalAveragePooling2D + Dense 120/66/66 (linear)
  softmax()          /root/reference/utils.py:7-11
  decode()           whenet.py:17-20, 28-33  expectation * 3 - 180 / - 99
  argmax             not in the reference; north-star's "bin argmax" = argmax of the logits
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

from . import b0_spec as G       # the oracle's own geometry (nothing is imported from the product package)

N_YAW, N_PITCH, N_ROLL = (n for _, n in G.BINS)


def normalise(img_u8: np.ndarray) -> np.ndarray:
    """whenet.py:23-26: float64 ``img/255`` then ``(img-mean)/std``; Keras then casts to
    float32 (whenet.py:27).  Returned as float32 -- the value the network is fed."""
    img = np.asarray(img_u8)
    if img.ndim != 4 or img.shape[1:] != (G.INPUT_SIZE, G.INPUT_SIZE, 3):
        raise ValueError(f"expected [N,224,224,3], got {img.shape}")
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]
    x = img / 255
    x = (x - mean) / std
    return x.astype(np.float32)


def normalise_lut() -> np.ndarray:
    """[3,256] float32: the exact image of normalise() for every uint8 value/channel."""
    lut = np.empty((3, 256), dtype=np.float32)
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]
    x = np.arange(256) / 255
    for c in range(3):
        lut[c] = ((x - mean[c]) / std[c]).astype(np.float32)
    return lut


def _pad_same(x: np.ndarray, k: int, s: int) -> Tuple[np.ndarray, int]:
    n, h, w, c = x.shape
    oh, pb, pa = G.tf_same(h, k, s)
    ow, qb, qa = G.tf_same(w, k, s)
    assert oh == ow
    xp = np.zeros((n, h + pb + pa, w + qb + qa, c), dtype=x.dtype)
    xp[:, pb:pb + h, qb:qb + w, :] = x
    return xp, oh


def conv2d(x: np.ndarray, w_hwio: np.ndarray, stride: int) -> np.ndarray:
    """TF 'SAME' Conv2D, NHWC, no bias."""
    kh, kw, cin, cout = w_hwio.shape
    if kh == 1 and stride == 1:
        return x @ w_hwio[0, 0].astype(x.dtype)
    xp, o = _pad_same(x, kh, stride)
    y = np.zeros((x.shape[0], o, o, cout), dtype=x.dtype)
    span = (o - 1) * stride + 1
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + span:stride, kx:kx + span:stride, :]
            y += patch @ w_hwio[ky, kx].astype(x.dtype)
    return y


def depthwise(x: np.ndarray, w_hwc1: np.ndarray, stride: int) -> np.ndarray:
    """TF 'SAME' DepthwiseConv2D (depth multiplier 1), NHWC, no bias."""
    kh, kw, c, _ = w_hwc1.shape
    xp, o = _pad_same(x, kh, stride)
    y = np.zeros((x.shape[0], o, o, c), dtype=x.dtype)
    span = (o - 1) * stride + 1
    for ky in range(kh):
        for kx in range(kw):
            y += xp[:, ky:ky + span:stride, kx:kx + span:stride, :] * w_hwc1[ky, kx, :, 0].astype(x.dtype)
    return y


def batchnorm(x: np.ndarray, w: Dict[str, np.ndarray], prefix: str) -> np.ndarray:
    dt = x.dtype
    g = w[f"{prefix}/gamma"].astype(dt)
    b = w[f"{prefix}/beta"].astype(dt)
    m = w[f"{prefix}/mean"].astype(dt)
    v = w[f"{prefix}/var"].astype(dt)
    return g * (x - m) / np.sqrt(v + dt.type(G.BN_EPSILON)) + b


def sigmoid(x: np.ndarray) -> np.ndarray:
    return 1 / (1 + np.exp(-x))


def swish(x: np.ndarray) -> np.ndarray:
    return x * sigmoid(x)


BNHook = Optional[Callable[[str, np.ndarray], None]]


def backbone(x: np.ndarray, w: Dict[str, np.ndarray], bn_hook: BNHook = None,
             taps: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
    """[N,224,224,3] normalised -> [N,7,7,1280].  ``bn_hook(prefix, bn_input)`` is called
    before every BN (used by the synthetic-weight calibration); ``taps`` collects named
    intermediate tensors for the per-kernel parity tests."""
    def bn(t: np.ndarray, prefix: str) -> np.ndarray:
        if bn_hook is not None:
            bn_hook(prefix, t)
        return batchnorm(t, w, prefix)

    def tap(name: str, t: np.ndarray) -> None:
        if taps is not None:
            taps[name] = t

    x = swish(bn(conv2d(x, w["stem/conv/kernel"], 2), "stem/bn"))
    tap("stem", x)
    for b in G.mbconv_blocks():
        p = f"b{b.number}"
        inp = x
        if b.expands:
            x = swish(bn(conv2d(x, w[f"{p}/expand/kernel"], 1), f"{p}/expand_bn"))
            tap(f"{p}/expand", x)
        x = swish(bn(depthwise(x, w[f"{p}/dw/kernel"], b.stride), f"{p}/dw_bn"))
        tap(f"{p}/dw", x)
        # SEBlock: mean over H,W (keepdims) -> conv1x1+bias -> swish -> conv1x1+bias -> sigmoid
        sq = x.mean(axis=(1, 2), keepdims=True)
        r = swish(sq @ w[f"{p}/se_reduce/kernel"][0, 0].astype(x.dtype) + w[f"{p}/se_reduce/bias"].astype(x.dtype))
        g = sigmoid(r @ w[f"{p}/se_expand/kernel"][0, 0].astype(x.dtype) + w[f"{p}/se_expand/bias"].astype(x.dtype))
        tap(f"{p}/gate", g)
        x = x * g
        x = bn(conv2d(x, w[f"{p}/project/kernel"], 1), f"{p}/project_bn")
        if b.identity_skip:
            x = x + inp
        tap(f"{p}/out", x)
    x = swish(bn(conv2d(x, w["head/conv/kernel"], 1), "head/bn"))
    tap("head", x)
    return x


def heads(feat: np.ndarray, w: Dict[str, np.ndarray]) -> np.ndarray:
    """GAP (whenet.py:10) + the three Dense heads (whenet.py:11-13) -> [N,252] logits in
    yaw|pitch|roll order."""
    f = feat.mean(axis=(1, 2))
    dt = f.dtype
    outs = [f @ w[f"{n}/kernel"].astype(dt) + w[f"{n}/bias"].astype(dt) for n in ("yaw", "pitch", "roll")]
    return np.concatenate(outs, axis=1)


def softmax(x: np.ndarray) -> np.ndarray:
    """utils.py:7-11 (the reference subtracts the row max in place)."""
    x = x - np.max(x, axis=1, keepdims=True)
    a = np.exp(x)
    b = np.sum(np.exp(x), axis=1, keepdims=True)
    return a / b


def decode(logits: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """whenet.py:28-33: softmax-expectation -> degrees."""
    dt = logits.dtype
    idx_yaw = np.arange(N_YAW, dtype=dt)
    idx = np.arange(N_PITCH, dtype=dt)
    ly = logits[:, :N_YAW]
    lp = logits[:, N_YAW:N_YAW + N_PITCH]
    lr = logits[:, N_YAW + N_PITCH:]
    yaw = np.sum(softmax(ly) * idx_yaw, axis=1) * 3 - 180
    pitch = np.sum(softmax(lp) * idx, axis=1) * 3 - 99
    roll = np.sum(softmax(lr) * idx, axis=1) * 3 - 99
    return yaw, pitch, roll


def argmax_bins(logits: np.ndarray) -> np.ndarray:
    """[N,3] int32: argmax of the yaw / pitch / roll logits (first maximum, as np.argmax)."""
    a = np.argmax(logits[:, :N_YAW], axis=1)
    b = np.argmax(logits[:, N_YAW:N_YAW + N_PITCH], axis=1)
    c = np.argmax(logits[:, N_YAW + N_PITCH:], axis=1)
    return np.stack([a, b, c], axis=1).astype(np.int32)


def top2_margin(logits: np.ndarray) -> np.ndarray:
    """[N,3]: gap between the two largest logits of each head (how fragile argmax is)."""
    out = []
    lo = 0
    for n in (N_YAW, N_PITCH, N_ROLL):
        s = np.sort(logits[:, lo:lo + n], axis=1)
        out.append(s[:, -1] - s[:, -2])
        lo += n
    return np.stack(out, axis=1)


def forward(img_u8: np.ndarray, w: Dict[str, np.ndarray], dtype=np.float64,
            taps: Optional[Dict[str, np.ndarray]] = None, chunk: int = 8) -> Dict[str, np.ndarray]:
    """Whole path: uint8 crops -> logits, angles (deg), argmax.  ``dtype`` is the arithmetic
    type of the network restatement (float64 = truth, float32 = noise-floor probe)."""
    x32 = normalise(img_u8)
    logits = []
    for i in range(0, x32.shape[0], chunk):          # whenet.py:27 batch_size=8 (numerically irrelevant)
        t = taps if (taps is not None and i == 0) else None
        f = backbone(x32[i:i + chunk].astype(dtype), w, taps=t)
        logits.append(heads(f, w))
    lg = np.concatenate(logits, axis=0)
    yaw, pitch, roll = decode(lg)
    return {"logits": lg, "yaw": yaw, "pitch": pitch, "roll": roll, "argmax": argmax_bins(lg)}


class OracleWHENet:
    """Same surface as whenet.WHENet (whenet.py:6-34), CPU float64, for tests."""

    def __init__(self, weights: Dict[str, np.ndarray], dtype=np.float64):
        self.w = weights
        self.dtype = dtype
        self.idx_tensor = np.arange(66, dtype=np.float32)
        self.idx_tensor_yaw = np.arange(120, dtype=np.float32)

    def get_angle(self, img):
        r = forward(np.asarray(img), self.w, self.dtype)
        return r["yaw"], r["pitch"], r["roll"]

    predict = get_angle
