"""Runs the REFERENCE'S OWN Python (/root/reference/*.py) as the checker.  TEST INFRASTRUCTURE.

The reference cannot be imported as it stands: its third-party runtime (tensorflow-gpu 1.12,
keras 2.1.6, efficientnet 0.0.4, opencv-python; /root/reference/requirements.txt:1-5) is not
installable here and the trained snapshots are absent (.MISSING_LARGE_BLOBS:1-2).  Everything the
reference wrote ITSELF on the hot path is plain Python/numpy, though, and runs once those imports
resolve.  This module puts stand-ins for the missing packages into ``sys.modules`` for the duration
of a ``with`` block and executes the reference's files from where they lie:

  whenet.py          WHENet.__init__ (graph wiring :8-14, load_weights :16, idx tensors :17-20) and
                     WHENet.get_angle (:22-34) run unmodified.  The ``keras`` stand-in is a tiny
                     symbolic-graph evaluator: it records the layers whenet.py wires up and, at
                     ``Model.predict(x, batch_size=8)``, casts to float32 (K.floatx), cuts the batch
                     into chunks of ``batch_size`` and evaluates GlobalAveragePooling2D / Dense itself.
                     ONLY the body of ``efn.EfficientNetB0`` -- third-party code the reference does
                     not contain -- comes from the restatement (oracle/whenet_oracle.backbone).
  utils.py           softmax (:7-11) and draw_axis (:13-46) run unmodified (``cv2`` stand-in records
                     the calls it receives).
  demo_video.py      process_detection (:11-35) runs unmodified on a frame object that records the
                     slice it is asked for.
  demo.py            its ``__main__`` body (:19-30) runs unmodified against the DROP-IN ``whenet`` module.
  yolo_v3/model.py   yolo_head, yolo_correct_boxes, yolo_boxes_and_scores, yolo_eval (:125-232) run
                     unmodified over a numpy-backed ``keras.backend`` / ``tensorflow`` stand-in (float32,
                     like the TF graph); only ``tf.image.non_max_suppression`` (TensorFlow's C++ kernel)
                     is the restatement (oracle/yolo_oracle.non_max_suppression).

Nothing here is imported by the product, by bench.py's timed region or by the -m gpu tests (the GPU
box has no /root/reference): the GPU side consumes the FIXTURES this harness produced
(tests/golden/make_reference_fixtures.py -> tests/golden/reference_*.npz).
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("WHENET_REFERENCE_DIR", "/root/reference")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "whenet.py"))


# --------------------------------------------------------------------------- keras / efficientnet
class Sym:
    """A symbolic tensor of the stand-in graph: (layer that produced it, its input)."""

    def __init__(self, layer, src=None):
        self.layer, self.src = layer, src


class _Layer:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs
        self.name = kwargs.get("name")

    def __call__(self, x):
        return Sym(self, x)


class GlobalAveragePooling2D(_Layer):
    def evaluate(self, x, w):
        return x.mean(axis=(1, 2))


class Dense(_Layer):
    def evaluate(self, x, w):
        key = self.name.split("_")[0]                  # 'yaw_new' -> the oracle's 'yaw/kernel', 'yaw/bias'
        k, b = w[f"{key}/kernel"].astype(x.dtype), w[f"{key}/bias"].astype(x.dtype)
        assert k.shape[1] == self.kwargs["units"], (self.name, k.shape)
        return x @ k + b


class _Backbone(_Layer):
    """The body of efn.EfficientNetB0(include_top=False): the only restated arithmetic."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def evaluate(self, x, w):
        return self.fn(x, w)


class _Input(_Layer):
    def evaluate(self, x, w):
        return x


class Model:
    """keras.models.Model(inputs=, outputs=[...]) with load_weights / predict / summary."""

    def __init__(self, inputs=None, outputs=None, weight_loader=None):
        self.inputs, self.outputs = inputs, list(outputs)
        self.weights = None
        self.predict_calls = []
        self._loader = weight_loader

    def load_weights(self, path):
        self.weights = _STATE["load_weights"](path)

    def summary(self):
        return "stand-in keras Model: " + ", ".join(o.layer.name or type(o.layer).__name__ for o in self.outputs)

    def _eval(self, sym, x, cache):
        if id(sym) in cache:
            return cache[id(sym)]
        v = x if sym.src is None else self._eval(sym.src, x, cache)
        out = sym.layer.evaluate(v, self.weights)
        cache[id(sym)] = out
        return out

    def predict(self, x, batch_size=32, **_):
        x = np.asarray(x)
        if x.ndim != 4 or x.shape[1:] != (224, 224, 3):
            raise ValueError("Error when checking : expected input_1 to have shape (None, 224, 224, 3) "
                             f"but got array with shape {x.shape}")
        x = x.astype(np.float32)                                   # Keras feeds K.floatx() = float32
        self.predict_calls.append((x.shape[0], batch_size))
        outs = [[] for _ in self.outputs]
        for i in range(0, x.shape[0], batch_size):                 # Keras' batch loop
            cache = {}
            chunk = x[i:i + batch_size].astype(_STATE["compute_dtype"])
            for j, o in enumerate(self.outputs):
                outs[j].append(self._eval(o, chunk, cache))
        res = []
        for j, o in enumerate(self.outputs):
            units = o.layer.kwargs.get("units", 0)
            arr = np.concatenate(outs[j], axis=0) if outs[j] else np.empty((0, units))
            res.append(arr.astype(np.float32))                     # Keras returns float32 arrays
        self.last_outputs = [a.copy() for a in res]                # utils.softmax works in place on what we return
        return res


_STATE = {"load_weights": None, "compute_dtype": np.float64}


def _keras_modules(backbone_fn):
    efn = types.ModuleType("efficientnet")

    def EfficientNetB0(include_top=True, input_shape=None, **kw):
        assert include_top is False and tuple(input_shape) == (224, 224, 3)
        inp = Sym(_Input())
        m = types.SimpleNamespace()
        m.input = inp
        m.output = Sym(_Backbone(backbone_fn), inp)
        return m

    efn.EfficientNetB0 = EfficientNetB0

    keras = types.ModuleType("keras")
    layers = types.ModuleType("keras.layers")
    layers.GlobalAveragePooling2D = GlobalAveragePooling2D
    layers.Dense = Dense
    for n in ("Conv2D", "Add", "ZeroPadding2D", "UpSampling2D", "Concatenate", "MaxPooling2D", "Input"):
        setattr(layers, n, type(n, (_Layer,), {}))
    adv = types.ModuleType("keras.layers.advanced_activations")
    adv.LeakyReLU = type("LeakyReLU", (_Layer,), {})
    norm = types.ModuleType("keras.layers.normalization")
    norm.BatchNormalization = type("BatchNormalization", (_Layer,), {})
    models = types.ModuleType("keras.models")
    models.Model = Model
    models.load_model = lambda *a, **k: (_ for _ in ()).throw(OSError("head_detect.h5 is not in the reference"))
    reg = types.ModuleType("keras.regularizers")
    reg.l2 = lambda v: ("l2", v)
    kutils = types.ModuleType("keras.utils")
    kutils.multi_gpu_model = lambda m, gpus=None: m
    keras.layers, keras.models, keras.regularizers, keras.utils = layers, models, reg, kutils
    keras.backend = _backend_module()
    layers.advanced_activations, layers.normalization = adv, norm
    return {"efficientnet": efn, "keras": keras, "keras.layers": layers, "keras.models": models,
            "keras.regularizers": reg, "keras.utils": kutils, "keras.backend": keras.backend,
            "keras.layers.advanced_activations": adv, "keras.layers.normalization": norm}


# --------------------------------------------------------------------------- keras.backend / tensorflow on numpy
def _backend_module():
    K = types.ModuleType("keras.backend")
    f32 = np.float32

    def constant(v, dtype=None, shape=None, name=None):
        return np.array(v, dtype=dtype or f32)

    def cast(x, dtype):
        return np.asarray(x).astype(dtype)

    def sigmoid(x):
        x = np.asarray(x, f32)
        return (f32(1) / (f32(1) + np.exp(-x))).astype(f32)

    K.reshape = lambda x, shape: np.reshape(x, [int(s) for s in shape])
    K.constant = constant
    K.shape = lambda x: np.array(np.shape(x), dtype=np.int32)
    K.arange = lambda start, stop=None, step=1, dtype="int32": np.arange(start, stop, step, dtype=dtype)
    K.tile = lambda x, n: np.tile(x, [int(v) for v in n])
    K.concatenate = lambda ts, axis=-1: np.concatenate(ts, axis=axis)
    K.cast = cast
    K.dtype = lambda x: np.asarray(x).dtype.name
    K.sigmoid = sigmoid
    K.exp = lambda x: np.exp(np.asarray(x, f32)).astype(f32)
    K.min = lambda x, axis=None, keepdims=False: np.min(x, axis=axis, keepdims=keepdims)
    K.round = np.round                                               # TF rounds half to even; so does numpy
    K.gather = lambda x, idx: np.asarray(x)[np.asarray(idx, dtype=np.int64)]
    K.ones_like = lambda x, dtype=None, name=None: np.ones_like(x, dtype=dtype)
    K.floatx = lambda: "float32"
    K.learning_phase = lambda: 0
    K.get_session = lambda: None
    K.placeholder = lambda shape=None, **kw: np.zeros(shape or ())
    return K


def _tensorflow_module(nms_fn):
    tf = types.ModuleType("tensorflow")
    tf.masked = []            # (tensor, mask) of every boolean_mask call: how a test sees yolo_eval's full decode

    def boolean_mask(x, mask):
        tf.masked.append((np.asarray(x), np.asarray(mask, dtype=bool)))
        return np.asarray(x)[np.asarray(mask, dtype=bool)]

    tf.boolean_mask = boolean_mask
    image = types.ModuleType("tensorflow.image")

    def non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5, **_):
        return np.asarray(nms_fn(np.asarray(boxes, np.float32), np.asarray(scores, np.float32),
                                 int(max_output_size), float(iou_threshold)), dtype=np.int64)

    image.non_max_suppression = non_max_suppression
    tf.image = image
    return {"tensorflow": tf, "tensorflow.image": image}


# --------------------------------------------------------------------------- cv2
class Cv2Recorder(types.ModuleType):
    """Stand-in for opencv-python: pixel functions come from PIL / the resize restatement, drawing and
    window functions only record their arguments (``.calls``)."""
    COLOR_BGR2RGB = 4
    FONT_HERSHEY_SIMPLEX = 0

    def __init__(self, resize_fn=None):
        super().__init__("cv2")
        self.calls = []
        self._resize = resize_fn

    def imread(self, path):
        from PIL import Image
        rgb = np.asarray(Image.open(path).convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])                # OpenCV decodes to BGR

    def cvtColor(self, img, code):
        assert code == self.COLOR_BGR2RGB
        return np.ascontiguousarray(np.asarray(img)[:, :, ::-1])

    def resize(self, img, dsize):
        self.calls.append(("resize", np.asarray(img).shape, tuple(dsize)))
        if self._resize is None:
            return np.zeros((dsize[1], dsize[0], 3), np.uint8)
        assert tuple(dsize) == (224, 224)
        return self._resize(np.asarray(img))

    def _record(name):                                               # noqa: N805
        def f(self, *a, **k):
            self.calls.append((name,) + tuple(x for x in a if not isinstance(x, (np.ndarray, RecordingFrame))))
            return 0 if name == "waitKey" else None
        return f

    rectangle = _record("rectangle")
    line = _record("line")
    putText = _record("putText")
    imshow = _record("imshow")
    waitKey = _record("waitKey")
    del _record


class RecordingFrame:
    """What demo_video.process_detection sees as ``img``: has ``.shape`` and records every slice."""

    def __init__(self, h, w):
        self.shape = (h, w, 3)
        self.slices = []

    def __getitem__(self, key):
        self.slices.append(key)
        return np.zeros((1, 1, 3), np.uint8)


# --------------------------------------------------------------------------- the context
@contextlib.contextmanager
def reference(backbone_fn=None, load_weights=None, compute_dtype=np.float64, nms_fn=None, resize_fn=None,
              dropin_whenet=False):
    """Installs the stand-ins, puts /root/reference first on sys.path, yields a namespace with ``cv2``
    (the recorder) and ``load(name)`` that imports a reference module FRESH; restores sys.modules and
    sys.path on exit.  ``dropin_whenet=True`` leaves module name ``whenet`` to the drop-in package (for
    demo.py, which imports it by that name)."""
    if not available():
        raise RuntimeError(f"{REF} is not present (build container only)")
    names = ["whenet", "utils", "demo", "demo_video", "yolo_v3", "yolo_v3.model", "yolo_v3.utils",
             "yolo_v3.yolo_postprocess", "cv2", "tensorflow", "tensorflow.image", "efficientnet"]
    mods = _keras_modules(backbone_fn or (lambda x, w: (_ for _ in ()).throw(RuntimeError("no backbone given"))))
    if nms_fn is None:
        def nms_fn(*a):
            raise RuntimeError("no NMS given")
    mods.update(_tensorflow_module(nms_fn))
    cv2 = Cv2Recorder(resize_fn)
    mods["cv2"] = cv2
    saved = {n: sys.modules.get(n) for n in set(names) | set(mods)}
    saved_path = list(sys.path)
    saved_state = dict(_STATE)
    _STATE["load_weights"] = load_weights
    _STATE["compute_dtype"] = compute_dtype
    try:
        for n in names:
            sys.modules.pop(n, None)
        sys.modules.update(mods)
        if dropin_whenet:
            pkg = os.path.join(ROOT, "headposeestimation-whenet_amd")
            sys.path[:] = [pkg, REF] + [p for p in saved_path if p not in (pkg, REF)]
        else:
            sys.path[:] = [REF] + [p for p in saved_path if p != REF]

        def load(name):
            m = importlib.import_module(name)
            f = getattr(m, "__file__", "") or ""
            if name != "whenet" or not dropin_whenet:
                assert os.path.abspath(f).startswith(os.path.abspath(REF)), (name, f)
            return m

        yield types.SimpleNamespace(cv2=cv2, tf=mods["tensorflow"], load=load, dir=REF)
    finally:
        sys.path[:] = saved_path
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
        _STATE.update(saved_state)


def run_get_angle(img, weights, backbone_fn, compute_dtype=np.float64):
    """The reference's WHENet('snapshot').get_angle(img), whenet.py:7-34 executed as written.
    Returns (yaw, pitch, roll, model) -- model.model.predict_calls shows the batch_size=8 chunking."""
    with reference(backbone_fn=backbone_fn, load_weights=lambda path: weights, compute_dtype=compute_dtype) as R:
        mod = R.load("whenet")
        m = mod.WHENet("WHENet.h5")
        y, p, r = m.get_angle(img)
        return y, p, r, m
