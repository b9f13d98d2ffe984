"""Packed WHENet weights ("WHNPACK1"): one flat file with the 315 Keras-native arrays.

The reference loads a Keras HDF5 snapshot (``model.load_weights(snapshot)``,
/root/reference/whenet.py:15-16).  ``WHENet.h5`` is not part of the reference tree
(.MISSING_LARGE_BLOBS:1), HDF5 cannot be read by the GPU-side C++ without libhdf5, and the
device wants BN-folded, fragment-ordered tensors anyway; so the snapshot format of this
framework is a flat little-endian container holding exactly the same arrays, in the same
(Keras) layouts, under canonical names (whenet_hip/spec.py::tensors).  BN folding and
device re-layout happen inside libwhenet_hip.so at load time (csrc/weights.cpp).

File layout (all little-endian):
    0   char[8]  "WHNPACK1"
    8   u32      version (1)
    12  u32      n_tensors
    16  u64      data_offset   (64-byte aligned, from file start)
    24  table: n_tensors x { u16 name_len, name, u8 dtype(0=f32), u8 ndim,
                             u32 dims[ndim], u64 offset (from data_offset), u64 nbytes }
    data_offset: tensor payloads, each 64-byte aligned.

Also here: the seeded synthetic snapshot used by tests and bench.py (the trained weights
are absent, so performance and parity are measured on random-init weights of the same
architecture -- said so wherever a number is reported).
"""
from __future__ import annotations

import hashlib
import io
import struct
from typing import Dict, Optional

import numpy as np

from . import spec

MAGIC = b"WHNPACK1"


def pack(weights: Dict[str, np.ndarray]) -> bytes:
    names = [t.name for t in spec.tensors()]
    missing = [n for n in names if n not in weights]
    if missing:
        raise ValueError(f"missing tensors: {missing[:4]}... ({len(missing)})")
    table = io.BytesIO()
    payload = io.BytesIO()
    for t in spec.tensors():
        a = np.ascontiguousarray(weights[t.name], dtype="<f4")
        if tuple(a.shape) != tuple(t.shape):
            raise ValueError(f"{t.name}: shape {a.shape} != expected {t.shape}")
        off = payload.tell()
        pad = (-off) % 64
        payload.write(b"\0" * pad)
        off += pad
        raw = a.tobytes()
        payload.write(raw)
        nb = t.name.encode()
        table.write(struct.pack("<H", len(nb)))
        table.write(nb)
        table.write(struct.pack("<BB", 0, a.ndim))
        table.write(struct.pack(f"<{a.ndim}I", *a.shape))
        table.write(struct.pack("<QQ", off, len(raw)))
    tb = table.getvalue()
    data_off = 24 + len(tb)
    data_off += (-data_off) % 64
    head = MAGIC + struct.pack("<IIQ", 1, len(names), data_off)
    blob = head + tb
    blob += b"\0" * (data_off - len(blob))
    return blob + payload.getvalue()


def unpack(blob: bytes) -> Dict[str, np.ndarray]:
    if blob[:8] != MAGIC:
        raise ValueError("not a WHNPACK1 file")
    ver, n, data_off = struct.unpack_from("<IIQ", blob, 8)
    if ver != 1:
        raise ValueError(f"unsupported WHNPACK version {ver}")
    p = 24
    out: Dict[str, np.ndarray] = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, p); p += 2
        name = blob[p:p + ln].decode(); p += ln
        dt, nd = struct.unpack_from("<BB", blob, p); p += 2
        dims = struct.unpack_from(f"<{nd}I", blob, p); p += 4 * nd
        off, nb = struct.unpack_from("<QQ", blob, p); p += 16
        if dt != 0:
            raise ValueError(f"{name}: unsupported dtype code {dt}")
        a = np.frombuffer(blob, dtype="<f4", count=nb // 4, offset=data_off + off)
        out[name] = a.reshape(dims).copy()
    return out


def save(path: str, weights: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(pack(weights))


def load(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        return unpack(f.read())


def is_packed(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except OSError:
        return False


def checksum(weights: Dict[str, np.ndarray]) -> str:
    h = hashlib.sha256()
    for t in spec.tensors():
        h.update(np.ascontiguousarray(weights[t.name], dtype="<f4").tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------------------
# Seeded synthetic snapshot
# --------------------------------------------------------------------------------------
def synthetic_raw(seed: int = 1234) -> Dict[str, np.ndarray]:
    """Random-init weights of the WHENet architecture, *before* BN calibration.

    Conv / depthwise kernels He-normal, BN gamma~U(0.5,1.5), beta~N(0,0.1),
    SE biases N(0,0.1).  BN moving statistics are placeholders (0 / 1) and the heads
    are zeros; :func:`apply_calibration` fills them.
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    skip_blocks = {f"b{b.index}" for b in spec.blocks() if b.has_skip}
    for t in spec.tensors():
        name, shape = t.name, t.shape
        leaf = name.rsplit("/", 1)[1]
        layer = name.rsplit("/", 1)[0]
        if leaf == "kernel" and len(shape) == 4:
            kh, kw, cin, cout = shape
            depthwise = layer.endswith("/dw")
            fan_in = kh * kw * (1 if depthwise else cin)
            gain = 2.0
            if layer.endswith("se_expand"):
                gain = 4.0
            a = rng.normal(0.0, np.sqrt(gain / fan_in), size=shape)
        elif leaf == "kernel":                      # dense heads: filled by calibration
            a = np.zeros(shape)
        elif leaf == "bias":
            if layer.endswith("se_expand"):
                a = rng.normal(0.0, 0.5, size=shape)
            elif layer.endswith("se_reduce"):
                a = rng.normal(0.0, 0.1, size=shape)
            else:
                a = np.zeros(shape)
        elif leaf == "gamma":
            a = rng.uniform(0.5, 1.5, size=shape)
            if layer.endswith("project_bn") and layer.split("/")[0] in skip_blocks:
                a *= 0.4       # residual branches contribute modestly, as in a trained net
        elif leaf == "beta":
            a = rng.normal(0.0, 0.1, size=shape)
        elif leaf == "mean":
            a = np.zeros(shape)
        elif leaf == "var":
            a = np.ones(shape)
        else:
            raise AssertionError(name)
        w[name] = a.astype(np.float32)
    return w


HEAD_SPREAD_BINS = 3.0      # softmax bump std, in bins (3 deg each): logits ~ -(j-mu)^2 / (2*3^2)
HEAD_CENTRE_OFFSET = {"yaw": -3.3, "pitch": 2.1, "roll": -1.7}


def head_basis(seed: int = 1234) -> Dict[str, Dict[str, np.ndarray]]:
    """Seeded directions the synthetic heads are built from (unit scale; gains come from
    the calibration fixture).  Per head:

      ``u``      (1280,) unit vector: the feature direction the pose angle reads out;
      ``smooth`` (1280,n) iid normal columns Gaussian-smoothed (sigma 3 bins) along the bin
                 axis -- circular for yaw (bins wrap at +-180 deg), reflected for pitch/roll --
                 unit std: a perturbation so the logits are not exactly rank-1.

    Why this shape: WHENet's heads are *ordinal* bin classifiers (3 deg per bin,
    whenet.py:11-13,31-33) trained so that the softmax is one narrow bump around the true
    angle; the expectation decode of such a bump is well conditioned.  iid-random Dense
    columns give several far-apart modes instead, where a 1e-4 logit perturbation (float32
    round-off through 82 conv layers) moves the expectation by centi-degrees -- an
    ill-conditioning the trained network does not have and the 1e-3 deg parity bar was not
    written for.  A linear head CAN emit an exact Gaussian bump: with
    mu = c + g*u.(f - fmean),  logit_j = (j'*mu' - j'^2/2)/s^2  (primes = minus n/2) equals
    -(j-mu)^2/(2 s^2) up to a per-row constant, linear in f.
    """
    rng = np.random.default_rng(seed + 1)
    taps = np.arange(-9, 10)
    gk = np.exp(-0.5 * (taps / 3.0) ** 2)
    gk /= gk.sum()
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for name, n in (("yaw", spec.N_YAW), ("pitch", spec.N_PITCH), ("roll", spec.N_ROLL)):
        u = rng.normal(0.0, 1.0, size=(spec.FEAT,))
        u /= np.linalg.norm(u)
        raw = rng.normal(0.0, 1.0, size=(spec.FEAT, n))
        mode = "wrap" if name == "yaw" else "reflect"
        padded = np.pad(raw, ((0, 0), (9, 9)), mode=mode)
        sm = np.zeros_like(raw)
        for t, gt in zip(taps, gk):
            sm += gt * padded[:, 9 + t: 9 + t + n]
        out[name] = {"u": u, "smooth": sm / sm.std()}
    return out


def apply_calibration(w: Dict[str, np.ndarray], calib: Dict[str, np.ndarray],
                      seed: int = 1234) -> Dict[str, np.ndarray]:
    """Install calibrated BN moving statistics and build the three Dense heads.

    ``calib`` holds ``<bn>/mean`` and ``<bn>/var`` for the 49 BN layers (the statistics the
    random-init network actually sees on the calibration crops -- what training would have
    left in the moving averages), ``feat_mean`` (1280,), and per head ``<head>/gain``
    (bins per unit of u.(f-fmean)) and ``<head>/pert`` (scale of the smooth perturbation).
    """
    out = dict(w)
    for bn in spec.bn_names():
        out[f"{bn}/mean"] = np.asarray(calib[f"{bn}/mean"], dtype=np.float32)
        out[f"{bn}/var"] = np.asarray(calib[f"{bn}/var"], dtype=np.float32)
    fmean = np.asarray(calib["feat_mean"], dtype=np.float64)
    basis = head_basis(seed)
    s2 = HEAD_SPREAD_BINS ** 2
    for name, n in (("yaw", spec.N_YAW), ("pitch", spec.N_PITCH), ("roll", spec.N_ROLL)):
        g = float(calib[f"{name}/gain"])
        pert = float(calib[f"{name}/pert"])
        jc = np.arange(n) - n / 2.0                                  # centred bin index j'
        k = np.outer(basis[name]["u"], g * jc / s2) + pert * basis[name]["smooth"]
        b = (jc * HEAD_CENTRE_OFFSET[name] - 0.5 * jc ** 2) / s2 - fmean @ k
        out[f"{name}/kernel"] = k.astype(np.float32)
        out[f"{name}/bias"] = b.astype(np.float32)
    return out


def synthetic(seed: int = 1234, calib_path: Optional[str] = None) -> Dict[str, np.ndarray]:
    """The synthetic snapshot every test / bench number in this repo is quoted on:
    ``synthetic_raw(seed)`` + the calibration data shipped inside the package
    (whenet_hip/data/calib_seed1234.npz, produced by tests/golden/make_golden.py)."""
    import os
    if calib_path is None:
        here = os.path.dirname(os.path.abspath(__file__))
        calib_path = os.path.join(here, "data", f"calib_seed{seed}.npz")
    with np.load(calib_path) as z:
        calib = {k: z[k] for k in z.files}
    return apply_calibration(synthetic_raw(seed), calib, seed)
