"""Keras-2.1.6 HDF5 snapshot  ->  WHNPACK1.

The reference constructs ``WHENet('WHENet.h5')`` and calls ``model.load_weights(snapshot)``
(/root/reference/whenet.py:15-16, demo.py:20): Keras' *topological* loader, which pairs the
file's weighted layers with the model's weighted layers **positionally** (file attr
``layer_names`` order == ``model.layers`` order of the saving model) and checks counts and
shapes.  This module does the same pairing against whenet_hip/spec.py::tensors() (creation
order of the 133 weighted layers / 315 arrays), so that the drop-in accepts the reference's own
snapshot file.  The three Dense heads are additionally matched by their explicit names
(``yaw_new`` / ``pitch_new`` / ``roll_new``, whenet.py:11-13).

h5py is not importable in the main interpreter of this image; when ``import h5py`` fails the
HDF5 file is read by a helper interpreter (``$WHENET_H5PY_PYTHON``, default
/opt/conda/bin/python3.9) running this very file in ``dump`` mode.  [RECOLLECTION of the Keras
file layout -- re-verify on the first real WHENet.h5: the trained file is absent from the
reference, .MISSING_LARGE_BLOBS:1.  Round 6: the reader accepts the layout variants the Keras versions around 2.1.6 write
(tests/test_keras_h5.py::test_layout_variants); every test file is still written by this module's own writer.]

CLI:  python keras_h5.py dump  <in.h5>  <out.npz>     (needs h5py)
      python keras_h5.py write <in.npz> <out.h5>      (needs h5py; test helper)
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
from typing import Dict, List, Tuple

import numpy as np

HELPER = os.environ.get("WHENET_H5PY_PYTHON", "/opt/conda/bin/python3.9")
Layer = Tuple[str, List[Tuple[str, np.ndarray]]]       # (layer name, [(weight name, array)])


# --------------------------------------------------------------------------------------------
# raw HDF5 access (h5py side; runs in whichever interpreter has h5py)
# --------------------------------------------------------------------------------------------
def _txt(n) -> str:
    return n.decode() if isinstance(n, bytes) else str(n)


def _dump_h5(path: str) -> List[Layer]:
    """Every layer of the file that holds weights, in ``layer_names`` order.  Accepts what the Keras versions the reference could
    have been saved with write: a weights-only file (``layer_names`` on the root) or a full-model file (``model_weights`` group,
    ``optimizer_weights`` beside it -- ignored); names as fixed-length bytes (h5py 2), variable-length bytes or str (h5py 3);
    weight names relative to the layer group with any number of path components (``kernel:0``, ``conv2d_1/kernel:0``,
    ``conv2d_1/conv2d_1/kernel:0``); layers without weights (activations, adds) interleaved."""
    import h5py
    out: List[Layer] = []
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f and "layer_names" not in f.attrs else f
        if "layer_names" not in g.attrs:
            raise ValueError(f"{path}: no 'layer_names' attribute on the root or on a 'model_weights' group: not a Keras weights file")
        names = [_txt(n) for n in np.asarray(g.attrs["layer_names"]).ravel()]
        for ln in names:
            if ln not in g:
                raise ValueError(f"{path}: layer_names lists '{ln}' but the file has no such group")
            lg = g[ln]
            wn = [_txt(n) for n in np.asarray(lg.attrs.get("weight_names", [])).ravel()]
            if len(wn):
                ws = []
                for n in wn:
                    if n not in lg:
                        raise ValueError(f"{path}: layer '{ln}': weight_names lists '{n}' but the group holds no such dataset")
                    ws.append((n, np.asarray(lg[n])))
                out.append((ln, ws))
    return out


def _write_h5(path: str, layers: List[Layer], opts: dict | None = None) -> None:
    """Test helper: the layer list as a Keras HDF5 file.  opts: full_model (model_weights + optimizer_weights groups), names
    ('fixed' bytes | 'vlen_bytes' | 'vlen_str'), weightless ([[position, layer name]]: groups with an empty weight_names)."""
    import h5py
    opts = opts or {}

    def names_attr(xs):
        kind = opts.get("names", "fixed")
        if kind == "fixed":
            return np.array([x.encode() for x in xs]) if xs else np.zeros((0,), "S1")
        dt = h5py.special_dtype(vlen=bytes if kind == "vlen_bytes" else str)
        return np.array([x.encode() if kind == "vlen_bytes" else x for x in xs], dtype=dt)

    with h5py.File(path, "w") as f:
        root = f.create_group("model_weights") if opts.get("full_model") else f
        order = [ln for ln, _ in layers]
        for pos, name in sorted(opts.get("weightless", []), key=lambda t: -t[0]):
            order.insert(pos, name)
        root.attrs["layer_names"] = names_attr(order)
        root.attrs["backend"] = b"tensorflow"
        root.attrs["keras_version"] = b"2.1.6"
        by = dict(layers)
        for ln in order:
            g = root.create_group(ln)
            ws = by.get(ln, [])
            g.attrs["weight_names"] = names_attr([n for n, _ in ws])
            for n, a in ws:
                g.create_dataset(n, data=np.asarray(a, np.float32))
        if opts.get("full_model"):
            f.attrs["keras_version"] = b"2.1.6"
            f.attrs["model_config"] = b'{"class_name": "Model"}'
            og = f.create_group("optimizer_weights")
            og.attrs["weight_names"] = names_attr(["Adam/iterations:0", "training/Adam/Variable:0"])
            og.create_dataset("Adam/iterations:0", data=np.zeros((), np.int64))
            og.create_dataset("training/Adam/Variable:0", data=np.zeros((3, 3, 3, 32), np.float32))


def _layers_to_npz(layers: List[Layer], path: str) -> None:
    d = {}
    for i, (ln, ws) in enumerate(layers):
        for j, (n, a) in enumerate(ws):
            d[f"{i:04d}|{j:02d}|{ln}|{n}"] = a
    np.savez(path, **d)


def _npz_to_layers(path: str) -> List[Layer]:
    layers: List[Layer] = []
    with np.load(path) as z:
        for key in sorted(z.files):
            i, _, ln, n = key.split("|", 3)
            if not layers or int(i) != len(layers) - 1:
                layers.append((ln, []))
            layers[-1][1].append((n, z[key]))
    return layers


def _helper(mode: str, src: str, dst: str) -> None:
    if not os.path.exists(HELPER):
        raise ImportError(f"h5py is not importable here and the helper interpreter {HELPER} does not exist; "
                          "set WHENET_H5PY_PYTHON or convert the snapshot elsewhere with tools/convert_h5.py")
    r = subprocess.run([HELPER, os.path.abspath(__file__), mode, src, dst], capture_output=True, text=True)
    if r.returncode != 0:
        last = r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""
        if last.startswith("ValueError: "):                  # a malformed file, as Keras' loader reports it
            raise ValueError(last[len("ValueError: "):])
        raise OSError(f"{HELPER} keras_h5.py {mode} failed:\n{r.stderr}")


def read_layers(path: str) -> List[Layer]:
    try:
        import h5py  # noqa: F401
        return _dump_h5(path)
    except ImportError:
        with tempfile.TemporaryDirectory() as td:
            npz = os.path.join(td, "dump.npz")
            _helper("dump", path, npz)
            return _npz_to_layers(npz)


def write_keras_h5(path: str, layers: List[Layer], **opts) -> None:
    try:
        import h5py  # noqa: F401
        _write_h5(path, layers, opts)
    except ImportError:
        import json
        with tempfile.TemporaryDirectory() as td:
            npz = os.path.join(td, "layers.npz")
            _layers_to_npz(layers, npz)
            with open(npz + ".opts.json", "w") as f:
                json.dump(opts, f)
            _helper("write", npz, path)


# --------------------------------------------------------------------------------------------
# pairing with the WHENet tensor census (main interpreter)
# --------------------------------------------------------------------------------------------
def _expected_layers():
    """[(canonical layer prefix, kind, [tensor names])] in creation order."""
    from . import spec
    out, cur = [], None
    for t in spec.tensors():
        prefix = t.name.rsplit("/", 1)[0]
        if cur is None or cur[0] != prefix:
            cur = (prefix, [])
            out.append(cur)
        cur[1].append(t)
    return out


_LEAF = {"kernel": "kernel", "depthwise_kernel": "kernel", "bias": "bias", "gamma": "gamma", "beta": "beta",
         "moving_mean": "mean", "moving_variance": "var"}


def _order_by_leaf(ln: str, ws, tensors):
    """The layer's arrays in the order of our tensors.  Keras pairs a layer's arrays positionally (layer.weights order); a
    file whose weight_names come in another order -- BatchNormalization's four arrays are the case that matters -- is
    re-ordered by the leaf of each name (``.../moving_mean:0`` -> mean) when every leaf is one Keras gives these layers."""
    leafs = [_LEAF.get(n.rsplit("/", 1)[-1].split(":")[0]) for n, _ in ws]
    want = [t.name.rsplit("/", 1)[1] for t in tensors]
    if None in leafs or len(set(leafs)) != len(leafs):
        return ws                                            # unknown names: positional, as Keras does
    if sorted(leafs) != sorted(want):
        raise ValueError(f"layer {ln}: arrays {[n for n, _ in ws]} do not match the expected {want}")
    by = dict(zip(leafs, ws))
    return [by[w] for w in want]


def convert_layers(layers: List[Layer]) -> Dict[str, np.ndarray]:
    """Positional (topological) pairing of the LAYERS, shape-checked; heads matched by name; a layer's arrays by leaf name."""
    expected = _expected_layers()
    heads = {"yaw": "yaw_new", "pitch": "pitch_new", "roll": "roll_new"}
    by_name = {ln: ws for ln, ws in layers}
    body = [(ln, ws) for ln, ws in layers if ln not in heads.values()]
    exp_body = [e for e in expected if e[0] not in heads]
    for hn in heads.values():
        if hn not in by_name:
            raise ValueError(f"snapshot has no layer named {hn} (whenet.py:11-13); its weighted layers are "
                             f"{[ln for ln, _ in layers][-4:]} at the end")
    if len(body) != len(exp_body):
        raise ValueError(f"You are trying to load a weight file containing {len(layers)} weighted layers into a "
                         f"model with {len(expected)} weighted layers (WHENet: 130 backbone + 3 Dense).")
    out: Dict[str, np.ndarray] = {}
    for (prefix, tensors), (ln, ws) in zip(exp_body, body):
        if len(ws) != len(tensors):
            raise ValueError(f"layer {ln} (-> {prefix}): {len(ws)} arrays in file, expected {len(tensors)}")
        ws = _order_by_leaf(ln, ws, tensors)
        for t, (wn, a) in zip(tensors, ws):
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"layer {ln}/{wn} (-> {t.name}): shape {tuple(a.shape)} != expected {t.shape}")
            out[t.name] = np.asarray(a, np.float32)
    for prefix, tensors in [e for e in expected if e[0] in heads]:
        ln = heads[prefix]
        if ln not in by_name:
            raise ValueError(f"snapshot has no layer named {ln} (whenet.py:11-13)")
        ws = by_name[ln]
        if len(ws) != len(tensors):
            raise ValueError(f"layer {ln}: {len(ws)} arrays in file, expected {len(tensors)}")
        ws = _order_by_leaf(ln, ws, tensors)
        for t, (wn, a) in zip(tensors, ws):
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"layer {ln}/{wn}: shape {tuple(a.shape)} != expected {t.shape}")
            out[t.name] = np.asarray(a, np.float32)
    return out


def convert(path: str) -> Dict[str, np.ndarray]:
    return convert_layers(read_layers(path))


def load_as_packed(path: str, cache: bool | None = None) -> bytes:
    """Keras .h5 -> WHNPACK1 bytes.

    Caching is OPT-IN (``cache=True`` or ``WHENET_H5_CACHE=1``): the converted snapshot is then kept
    next to the file as ``<path>.whnp`` together with ``<path>.whnp.sha256`` = SHA-256 of the HDF5
    file it was converted from; the cache is used only while that digest matches the file's current
    content (a replaced WHENet.h5 -- whatever its mtime -- or a stray .whnp is never trusted)."""
    import hashlib
    from . import weights as W
    if cache is None:
        cache = os.environ.get("WHENET_H5_CACHE", "0") not in ("", "0")
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != b"\x89HDF\r\n\x1a\n":
        raise ValueError(f"{path}: neither a WHNPACK1 snapshot nor an HDF5 file")
    digest = hashlib.sha256(raw).hexdigest()
    cpath, kpath = path + ".whnp", path + ".whnp.sha256"
    if cache and os.path.exists(cpath) and os.path.exists(kpath):
        try:
            with open(kpath) as f:
                key = f.read().strip()
            if key == digest and W.is_packed(cpath):
                with open(cpath, "rb") as f:
                    return f.read()
        except OSError:
            pass
    blob = W.pack(convert(path))
    if cache:
        try:
            with open(cpath, "wb") as f:
                f.write(blob)
            with open(kpath, "w") as f:
                f.write(digest + "\n")
        except OSError:
            pass
    return blob


def to_keras_layers(weights: Dict[str, np.ndarray], offset: int = 0, style: str = "plain") -> List[Layer]:
    """Our tensors -> the layer list a Keras 2.1.6 save of the reference model would hold
    (auto-generated names conv2d_N / batch_normalization_N / depthwise_conv2d_N numbered in
    creation order from 1+offset; heads by their explicit names).  Test helper."""
    counters = {"conv2d": offset, "batch_normalization": offset, "depthwise_conv2d": offset}
    wname = {"kernel": "kernel:0", "bias": "bias:0", "gamma": "gamma:0", "beta": "beta:0",
             "mean": "moving_mean:0", "var": "moving_variance:0"}
    layers: List[Layer] = []
    for prefix, tensors in _expected_layers():
        leafs = [t.name.rsplit("/", 1)[1] for t in tensors]
        if prefix in ("yaw", "pitch", "roll"):
            ln = prefix + "_new"
        elif "gamma" in leafs:
            counters["batch_normalization"] += 1
            ln = f"batch_normalization_{counters['batch_normalization']}"
        elif prefix.endswith("/dw"):
            counters["depthwise_conv2d"] += 1
            ln = f"depthwise_conv2d_{counters['depthwise_conv2d']}"
        else:
            counters["conv2d"] += 1
            ln = f"conv2d_{counters['conv2d']}"
        ws = []
        for t, leaf in zip(tensors, leafs):
            n = wname[leaf]
            if prefix.endswith("/dw") and leaf == "kernel":
                n = "depthwise_kernel:0"
            # weight names relative to the layer group: Keras 2.1.6 'conv2d_1/kernel:0'; 'double' = with the layer's name scope
            # twice (later Keras / nested models), 'bare' = without any
            full = {"plain": f"{ln}/{n}", "double": f"{ln}/{ln}/{n}", "bare": n}[style]
            ws.append((full, weights[t.name]))
        layers.append((ln, ws))
    return layers


if __name__ == "__main__":
    mode, src, dst = sys.argv[1:4]
    if mode == "dump":
        _layers_to_npz(_dump_h5(src), dst)
    elif mode == "write":
        import json
        o = {}
        if os.path.exists(src + ".opts.json"):
            with open(src + ".opts.json") as fo:
                o = json.load(fo)
        _write_h5(dst, _npz_to_layers(src), o)
    else:
        raise SystemExit(f"unknown mode {mode}")
