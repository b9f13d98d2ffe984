"""Drop-in replacement for the reference module ``whenet`` (/root/reference/whenet.py).

    from whenet import WHENet            # demo.py:3, demo_video.py:3
    model = WHENet('WHENet.h5')          # demo.py:20     (positional snapshot path)
    model = WHENet(snapshot=path)        # demo_video.py:40
    print(model.model.summary())         # demo.py:22
    yaw, pitch, roll = model.get_angle(img_rgb_uint8[N,224,224,3])     # demo.py:14, demo_video.py:27

Same class name, constructor, method names, argument meaning, return types (three float32
arrays of shape (N,)) and error classes (ValueError for a bad input shape, OSError for a
missing snapshot) as the reference.  Behind it, instead of Keras/TensorFlow, is
libwhenet_hip.so: hand-written gfx950 kernels reached through a C ABI (include/whenet_hip.h)
via ctypes.  There is no NumPy/CPU implementation in this module: without the built library
and an MI355X the constructor raises.

Extras that the reference does not have (keyword-only, all optional):
  device=0            GPU ordinal
  devices=[0,1,..]    several GPUs from this one process: one handle + feeder thread per device, the batch of each
                      get_angle call split contiguously over them (whenet_hip/multi.py), same bits
  dtype='f32s'|'f32'|'f16'   activation / 1x1-weight type.  Default (round 6) 'f32s': float32 storage with the 1x1 products as binary16
                      hi/lo pairs on the f16 matrix cores (include/whenet_hip.h WHENET_F32S) -- the same 1e-3 degree / equal-argmax bar
                      as 'f32' (the exact-float32 configuration) at 1.37 x its throughput and 0.9 x its batch-1 latency.  Its
                      precondition is activations inside binary16's range (|x| <= 65504): should a snapshot break it the angles
                      come out NaN, and this class then switches the handle to the exact-f32 kernels (option split_pw = 0) and
                      repeats the call -- a warning says so
  inflight=1..4       engines of the handle for callers that keep several forwards in flight themselves (whenet_forward_u8_device /
                      submit); one large get_angle call (N >= 256) is cut into 128-crop chunks over the library's own fan-out engines
                      whatever this is (include/whenet_hip.h "fanout_min", "fanout_engines")
  .predict            alias of get_angle (BASELINE.json words the API as WHENet.predict(crop))
  .last_logits, .last_argmax   what Model.predict returned for the last call, and the bin argmax
"""
from __future__ import annotations

import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from whenet_hip import _lib, spec  # noqa: E402
from whenet_hip import weights as _weights  # noqa: E402

_DTYPES = {"f32": _lib.F32, "fp32": _lib.F32, "float32": _lib.F32,
           "f16": _lib.F16, "fp16": _lib.F16, "float16": _lib.F16,
           "f32s": _lib.F32S}          # float32 storage, 1x1 products as binary16 hi/lo pairs on the f16 matrix cores


_as_uint8_crops = _lib.as_uint8_crops


def _is_byte_valued(a: np.ndarray) -> bool:
    if a.dtype == np.uint8:
        return True
    return bool(a.size == 0 or (a.min() >= 0 and a.max() <= 255 and np.all(a == np.rint(a))))


def _check_shape(a: np.ndarray) -> None:
    if a.ndim != 4 or tuple(a.shape[1:]) != (spec.IMG, spec.IMG, 3):
        raise ValueError(f"Error when checking input: expected input to have shape "
                         f"(None, 224, 224, 3) but got array with shape {a.shape}")


class _Model:
    """Stand-in for the inner ``keras.models.Model`` (whenet.py:14): ``summary()`` (demo.py:22)
    and ``predict()`` (whenet.py:27)."""

    def __init__(self, outer: "WHENet"):
        self._outer = outer

    def summary(self):
        h = self._outer._handle
        info = h.info()
        print("_" * 78)
        print(f"{'Layer (kernel)':34s}{'Output shape':22s}{'Param #':>12s}")
        print("=" * 78)
        shapes = {t.name: t.shape for t in spec.tensors()}

        def count(prefixes):
            return sum(int(np.prod(s)) for n, s in shapes.items() if any(n.startswith(p + "/") for p in prefixes))

        print(f"{'stem conv3x3/s2+BN+swish':34s}{'(None, 112, 112, 32)':22s}{count(['stem']):>12,d}")
        for b in spec.blocks():
            p = f"b{b.index}"
            desc = f"{p} MBConv{b.expand} k{b.k} s{b.s}" + (" +skip" if b.has_skip else "")
            print(f"{desc:34s}{str((None, b.h_out, b.h_out, b.cout)):22s}{count([p]):>12,d}")
        print(f"{'head conv1x1+BN+swish':34s}{'(None, 7, 7, 1280)':22s}{count(['head']):>12,d}")
        print(f"{'global_average_pooling2d':34s}{'(None, 1280)':22s}{0:>12,d}")
        for n, k in (("yaw_new", 120), ("pitch_new", 66), ("roll_new", 66)):
            print(f"{n + ' (Dense)':34s}{str((None, k)):22s}{count([n.split('_')[0]]):>12,d}")
        print("=" * 78)
        total = info.params_backbone + info.params_heads
        print(f"Total params: {total:,d}")
        print(f"Backend: libwhenet_hip on {info.device_name.decode()} ({info.arch.decode()}), "
              f"dtype {'f16' if info.dtype == _lib.F16 else ('f32s' if self._outer._dtype == _lib.F32S else 'f32')}, "
              f"{info.n_kernels_per_forward} kernels/forward")
        print("_" * 78)

    def predict(self, x, batch_size=8, **_):
        """Model.predict(img, batch_size=8) (whenet.py:27): ``x`` is the *normalised* image the
        reference passes; Keras casts it to float32 and returns the three logit arrays.
        ``batch_size`` only caps Keras' internal chunking (numerically irrelevant)."""
        x = np.asarray(x)
        _check_shape(x)
        if x.shape[0] == 0:
            return [np.empty((0, 120), np.float32), np.empty((0, 66), np.float32), np.empty((0, 66), np.float32)]
        _, _, lg = self._outer._forward_f32(np.ascontiguousarray(x, dtype=np.float32))
        return [lg[:, :120].copy(), lg[:, 120:186].copy(), lg[:, 186:].copy()]


class WHENet:
    def __init__(self, snapshot=None, *, device=0, dtype="f32s", devices=None, inflight=None):
        if isinstance(dtype, str):
            if dtype.lower() not in _DTYPES:
                raise ValueError(f"dtype must be one of {sorted(set(_DTYPES))}")
            dtype = _DTYPES[dtype.lower()]
        if snapshot is None:
            # whenet.py:15: no snapshot -> the freshly initialised network.  Ours: the seeded
            # random-init snapshot (whenet_hip/weights.py::synthetic).
            snapshot = _weights.pack(_weights.synthetic(1234))
        elif isinstance(snapshot, (str, os.PathLike)):
            path = os.fspath(snapshot)
            if not os.path.exists(path):
                raise OSError(f"Unable to open file (unable to open file: name = '{path}')")
            if not _weights.is_packed(path):
                from whenet_hip import keras_h5
                snapshot = keras_h5.load_as_packed(path)          # Keras HDF5 -> WHNPACK1 bytes
            else:
                snapshot = path
        if inflight is not None and not (1 <= int(inflight) <= 4):
            raise ValueError("inflight must be 1..4")
        if devices is not None:
            from whenet_hip.multi import MultiDeviceHandle
            self._handle = MultiDeviceHandle(snapshot, [int(d) for d in devices], dtype)
        else:
            self._handle = _lib.Handle(snapshot, device=int(device), dtype=dtype)
        self._dtype = dtype
        if inflight is not None:
            self._handle.set_option("inflight", int(inflight))
        self.model = _Model(self)
        self.idx_tensor = [idx for idx in range(66)]                       # whenet.py:17-20
        self.idx_tensor = np.array(self.idx_tensor, dtype=np.float32)
        self.idx_tensor_yaw = [idx for idx in range(120)]
        self.idx_tensor_yaw = np.array(self.idx_tensor_yaw, dtype=np.float32)
        self.last_logits = None
        self.last_argmax = None

    FANOUT_MIN = 256          # include/whenet_hip.h "fanout_min": the library cuts larger blocking calls into chunks over its own
                              # fan-out engines (created on the first such call; "inflight" and the other entry points are not touched)

    def _out_of_range(self, ypr) -> bool:
        """WHENET_F32S splits activations into binary16 hi/lo halves: beyond |x| = 65504 the hi half is inf and the angles NaN (never a
        silently wrong number).  Not reachable with EfficientNet-B0 behind BatchNorm in practice; if a snapshot does it, the handle
        goes over to the exact-float32 kernels for good (option split_pw = 0: bitwise a WHENET_F32 handle)."""
        if self._dtype != _lib.F32S or np.isfinite(ypr).all():
            return False
        import warnings
        warnings.warn("whenet: activations outside binary16's range with dtype='f32s'; this model now runs the exact-float32 kernels "
                      "(option split_pw=0)", RuntimeWarning, stacklevel=3)
        self._handle.set_option("split_pw", 0)
        self._dtype = _lib.F32
        return True

    def _forward(self, u8: np.ndarray):
        ypr, am, lg = self._handle.forward(u8, want_logits=True)
        if self._out_of_range(ypr):
            ypr, am, lg = self._handle.forward(u8, want_logits=True)
        self.last_logits, self.last_argmax = lg, am
        return ypr, am, lg

    def _forward_f32(self, x: np.ndarray):
        ypr, am, lg = self._handle.forward_f32(x, want_logits=True)
        if self._out_of_range(ypr):
            ypr, am, lg = self._handle.forward_f32(x, want_logits=True)
        self.last_logits, self.last_argmax = lg, am
        return ypr, am, lg

    def get_angle(self, img):
        """whenet.py:22-34.  img: [N,224,224,3] RGB -> (yaw, pitch, roll), float32 (N,).

        8-bit crops (uint8, or any numeric dtype holding integers 0..255 -- what cv2.resize of an
        image gives, demo.py:11) go to the device as bytes and are normalised there through a LUT that
        is the exact image of whenet.py:23-26.  Anything else numeric is handled as the reference
        handles it: ``img/255`` and ``(img-mean)/std`` in float64 (whenet.py:23-26), cast to float32
        as Keras does at Model.predict (whenet.py:27), then the same network."""
        a = np.asarray(img)
        _check_shape(a)
        if a.dtype == object or not (np.issubdtype(a.dtype, np.number) or a.dtype == np.bool_):
            raise ValueError(f"get_angle: crops must be numeric, got dtype {a.dtype}")
        if a.shape[0] == 0:
            e = np.empty((0,), np.float32)
            return e, e.copy(), e.copy()
        if _is_byte_valued(a):
            ypr, _, _ = self._forward(np.ascontiguousarray(a.astype(np.uint8, copy=False)))
        else:
            x = a / 255                                                   # whenet.py:25
            x = (x - list(spec.MEAN)) / list(spec.STD)                    # whenet.py:26
            ypr, _, _ = self._forward_f32(np.ascontiguousarray(x, dtype=np.float32))
        return ypr[:, 0].copy(), ypr[:, 1].copy(), ypr[:, 2].copy()

    predict = get_angle

    def close(self):
        if getattr(self, "_handle", None) is not None:
            self._handle.close()
            self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
