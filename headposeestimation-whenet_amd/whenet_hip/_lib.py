"""ctypes binding of libwhenet_hip.so (include/whenet_hip.h).  Thin: argument marshalling
and error-code -> exception mapping only.  There is no Python/NumPy implementation of the
path behind it -- if the library or a gfx950 device is missing, callers get an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

ABI_VERSION = 4
F32, F16, F32S = 0, 1, 2          # include/whenet_hip.h WHENET_F32 / WHENET_F16 / WHENET_F32S
OK, ENOENT, EIO, ENOMEM, ENODEV, EINVAL, EFORMAT, EHIP = 0, -2, -5, -12, -19, -22, -74, -1000
MAX_INFLIGHT = 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "lib", "libwhenet_hip.so")


class Info(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("dtype", C.c_int32), ("device_id", C.c_int32),
                ("compute_units", C.c_int32), ("params_backbone", C.c_int64),
                ("params_heads", C.c_int64), ("n_tensors", C.c_int32),
                ("n_kernels_per_forward", C.c_int32), ("macs_per_crop", C.c_int64),
                ("arena_bytes", C.c_int64), ("capacity", C.c_int32), ("graph_enabled", C.c_int32),
                ("device_name", C.c_char * 64), ("arch", C.c_char * 32)]


class LaunchStat(C.Structure):
    _fields_ = [("layer", C.c_char * 32), ("kind", C.c_char * 16), ("kernel", C.c_char * 64),
                ("avg_us", C.c_double), ("alg_bytes", C.c_double), ("alg_flops", C.c_double),
                ("crops", C.c_int32), ("chains", C.c_int32)]


_P = C.c_void_p
_PROTOS = {
    "whenet_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "whenet_create_from_memory": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.POINTER(_P)]),
    "whenet_create_postproc": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "whenet_destroy": (None, [_P]),
    "whenet_last_error": (C.c_char_p, [_P]),
    "whenet_get_info": (C.c_int, [_P, C.POINTER(Info)]),
    "whenet_set_option": (C.c_int, [_P, C.c_char_p, C.c_long]),
    "whenet_forward_u8": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "whenet_forward_f32": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "whenet_forward_u8_device": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    "whenet_sync": (C.c_int, [_P]),
    "whenet_submit_u8": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "whenet_collect": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "whenet_frame_rects": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P]),
    "whenet_normalise_table": (C.c_int, [_P]),
    "whenet_submit_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "whenet_op_crop_resize": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "whenet_yolo_eval": (C.c_int, [_P, C.POINTER(_P), _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int, _P, _P, _P, _P, C.POINTER(C.c_int), _P, _P]),
    "whenet_profile": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(LaunchStat), C.c_int, C.POINTER(C.c_int)]),
    "whenet_op_stem": (C.c_int, [_P, _P, C.c_int, _P]),
    "whenet_op_block": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "whenet_op_block_range": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P]),
    "whenet_op_head": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    "whenet_op_decode": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "whenet_block_spec": (C.c_int, [C.c_int, C.POINTER(C.c_int32 * 8)]),
    "whenet_dw_plan": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32 * 12)]),
    "whenet_front_plan": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32 * 12)]),
    "whenet_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "whenet_device_free": (C.c_int, [_P, _P]),
    "whenet_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "whenet_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
}
EXPORTS = tuple(_PROTOS)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the library and bind every symbol the header declares (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("WHENET_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build it with `python __graft_entry__.py` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


RGB, BGR = 0, 1


def as_uint8_crops(img) -> np.ndarray:
    """Validate like Keras would at Model.predict (ValueError on wrong rank/shape) and return a
    contiguous uint8 array.  The reference divides by 255 whatever the dtype
    (/root/reference/whenet.py:25); integer-valued arrays in [0,255] of any dtype are therefore
    accepted here.  (Real-valued crops take WHENet.get_angle's float path instead.)"""
    a = np.asarray(img)
    if a.ndim != 4 or tuple(a.shape[1:]) != (224, 224, 3):
        raise ValueError(f"Error when checking input: expected input to have shape "
                         f"(None, 224, 224, 3) but got array with shape {a.shape}")
    if a.dtype != np.uint8:
        if a.dtype == object or not np.issubdtype(a.dtype, np.number):
            raise ValueError(f"crops must be numeric, got dtype {a.dtype}")
        if a.size and (a.min() < 0 or a.max() > 255 or not np.all(a == np.rint(a))):
            raise ValueError("not 8-bit RGB crops (integer values 0..255), as produced by cv2.resize on an "
                             "image (demo.py:11)")
        a = a.astype(np.uint8)
    return np.ascontiguousarray(a)


def _frame_u8(frame) -> np.ndarray:
    frame = np.asarray(frame)
    if frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8:
        raise ValueError(f"frame must be uint8 [H,W,3], got {frame.dtype} {frame.shape}")
    return np.ascontiguousarray(frame)


def frame_rects(frame_h: int, frame_w: int, bboxes) -> np.ndarray:
    """demo_video.py:13-21 for k YOLO boxes (y_min, x_min, y_max, x_max) -> int32 [k,4] windows
    (y0, x0, y1, x1).  Pure host arithmetic inside the library (no GPU needed)."""
    b = np.ascontiguousarray(bboxes, np.float32).reshape(-1, 4)
    out = np.empty((b.shape[0], 4), np.int32)
    code = load().whenet_frame_rects(int(frame_h), int(frame_w), _ptr(b), b.shape[0], _ptr(out))
    raise_for(code, "whenet_frame_rects: bad arguments")
    return out


def normalise_lut() -> np.ndarray:
    """[3,256] float32: whenet.py:23-26 + Keras' float32 cast for every byte value, as the stem kernels apply it.
    Pure host arithmetic inside the library (no GPU needed)."""
    out = np.empty((3, 256), np.float32)
    raise_for(load().whenet_normalise_table(_ptr(out)), "whenet_normalise_table: bad arguments")
    return out


class WhenetError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libwhenet_hip error {code}: {msg}")
        self.code = code


def raise_for(code: int, msg: str):
    """Map ABI codes onto what the reference's Keras path raises:
    missing/unreadable snapshot -> OSError; bad shape/argument/format -> ValueError."""
    if code == OK:
        return
    if code in (ENOENT, EIO):
        raise OSError(msg)
    if code in (EINVAL, EFORMAT):
        raise ValueError(msg)
    if code == ENOMEM:
        raise MemoryError(msg)
    raise WhenetError(code, msg)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_P)


class Handle:
    """Owns one whenet_t*."""

    def __init__(self, snapshot, device: int = 0, dtype: int = F32):
        lib = load()
        h = _P()
        if isinstance(snapshot, (bytes, bytearray, memoryview)):
            buf = (C.c_char * len(snapshot)).from_buffer_copy(bytes(snapshot))
            rc = lib.whenet_create_from_memory(C.cast(buf, _P), len(snapshot), device, dtype, C.byref(h))
        else:
            rc = lib.whenet_create(os.fsencode(snapshot), device, dtype, C.byref(h))
        if rc != OK:
            raise_for(rc, (lib.whenet_last_error(None) or b"").decode(errors="replace"))
        self._h = h
        self._lib = lib
        self.dtype = dtype
        self.device = device

    @classmethod
    def postproc(cls, device: int = 0) -> "Handle":
        """A handle WITHOUT a network (whenet_create_postproc): device, stream and scratch for the frame / detector
        stages (yolo_eval, op_crop_resize); forwards raise."""
        lib = load()
        h = _P()
        rc = lib.whenet_create_postproc(device, C.byref(h))
        if rc != OK:
            raise_for(rc, (lib.whenet_last_error(None) or b"").decode(errors="replace"))
        self = cls.__new__(cls)
        self._h = h
        self._lib = lib
        self.dtype = F32
        self.device = device
        return self

    def _check(self, rc: int):
        if rc != OK:
            raise_for(rc, (self._lib.whenet_last_error(self._h) or b"").decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.whenet_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path ---------------------------------------------------------------------
    def forward(self, crops: np.ndarray, want_logits: bool = True):
        # the C side reads n*150,528 bytes from this pointer: never hand it anything else
        if not (isinstance(crops, np.ndarray) and crops.dtype == np.uint8 and crops.ndim == 4
                and crops.shape[1:] == (224, 224, 3) and crops.flags.c_contiguous):
            raise ValueError("Handle.forward needs a C-contiguous uint8 array [n,224,224,3] "
                             f"(got {getattr(crops, 'dtype', type(crops))} {getattr(crops, 'shape', '')}); "
                             "use whenet_hip._lib.as_uint8_crops()")
        n = crops.shape[0]
        if n == 0:
            return (np.empty((0, 3), np.float32), np.empty((0, 3), np.int32),
                    np.empty((0, 252), np.float32) if want_logits else None)
        ypr = np.empty((n, 3), np.float32)
        am = np.empty((n, 3), np.int32)
        lg = np.empty((n, 252), np.float32) if want_logits else None
        self._check(self._lib.whenet_forward_u8(self._h, _ptr(crops), n, _ptr(ypr), _ptr(am), _ptr(lg)))
        return ypr, am, lg

    def forward_f32(self, x: np.ndarray, want_logits: bool = True):
        """x: the NORMALISED float32 image [n,224,224,3] (what whenet.py:27 feeds Model.predict)."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.ndim == 4
                and x.shape[1:] == (224, 224, 3) and x.flags.c_contiguous and x.shape[0] >= 1):
            raise ValueError("Handle.forward_f32 needs a C-contiguous float32 array [n>=1,224,224,3]")
        n = x.shape[0]
        ypr = np.empty((n, 3), np.float32)
        am = np.empty((n, 3), np.int32)
        lg = np.empty((n, 252), np.float32) if want_logits else None
        self._check(self._lib.whenet_forward_f32(self._h, _ptr(x), n, _ptr(ypr), _ptr(am), _ptr(lg)))
        return ypr, am, lg

    def forward_device(self, d_crops: int, n: int, d_ypr: int, d_argmax: int = 0, d_logits: int = 0, stream: int = 0):
        self._check(self._lib.whenet_forward_u8_device(self._h, d_crops, n, d_ypr, d_argmax or None,
                                                       d_logits or None, stream or None))

    def sync(self):
        self._check(self._lib.whenet_sync(self._h))

    def submit(self, crops: np.ndarray) -> int:
        if not (isinstance(crops, np.ndarray) and crops.dtype == np.uint8 and crops.ndim == 4
                and crops.shape[1:] == (224, 224, 3) and crops.flags.c_contiguous and crops.shape[0] >= 1):
            raise ValueError("Handle.submit needs a C-contiguous uint8 array [n>=1,224,224,3]")
        t = C.c_int(-1)
        self._check(self._lib.whenet_submit_u8(self._h, _ptr(crops), crops.shape[0], C.byref(t)))
        return t.value

    def submit_frame(self, frame: np.ndarray, rects: np.ndarray, bgr: bool = True) -> int:
        """frame uint8 [H,W,3]; rects int32 [k,4] (y0,x0,y1,x1) -> ticket (collect with n=k)."""
        frame = _frame_u8(frame)
        rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
        t = C.c_int(-1)
        self._check(self._lib.whenet_submit_frame(self._h, _ptr(frame), frame.shape[0], frame.shape[1],
                                                  BGR if bgr else RGB, _ptr(rects), rects.shape[0], C.byref(t)))
        return t.value

    def op_crop_resize(self, frame: np.ndarray, rects: np.ndarray, bgr: bool = True) -> np.ndarray:
        frame = _frame_u8(frame)
        rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
        out = np.empty((rects.shape[0], 224, 224, 3), np.uint8)
        self._check(self._lib.whenet_op_crop_resize(self._h, _ptr(frame), frame.shape[0], frame.shape[1],
                                                    BGR if bgr else RGB, _ptr(rects), rects.shape[0], _ptr(out)))
        return out

    def yolo_eval(self, yolo_outputs, anchors, num_classes: int, image_shape, max_boxes: int = 20,
                  score_threshold: float = .6, iou_threshold: float = .5, debug: bool = False):
        """yolo_v3/model.py:193-232 on numpy feature maps [gh, gw, 3*(5+C)] (or with a leading batch axis of 1).
        Returns boxes [k,4] (y_min, x_min, y_max, x_max), scores [k], classes [k]; with debug also the box
        indices and every decoded box / score."""
        maps = []
        for m in yolo_outputs:
            m = np.asarray(m)
            if m.ndim == 4:
                if m.shape[0] != 1:
                    raise ValueError("yolo_eval: batch of one, as YOLO.detect feeds it")
                m = m[0]
            if m.ndim != 3 or m.shape[2] != 3 * (5 + num_classes):
                raise ValueError(f"yolo_eval: feature map of shape {m.shape}, expected [gh, gw, {3 * (5 + num_classes)}]")
            maps.append(np.ascontiguousarray(m, np.float32))
        anchors = np.ascontiguousarray(anchors, np.float32).reshape(-1, 2)
        L = len(maps)
        ptrs = (_P * L)(*[_ptr(m) for m in maps])
        gh = np.array([m.shape[0] for m in maps], np.int32)
        gw = np.array([m.shape[1] for m in maps], np.int32)
        n_all = int(sum(m.shape[0] * m.shape[1] * 3 for m in maps))
        if max_boxes < 1:
            raise ValueError("yolo_eval: max_boxes must be >= 1")
        cap = num_classes * min(int(max_boxes), n_all)          # (no more selections per class than boxes)
        boxes = np.empty((cap, 4), np.float32)
        scores = np.empty(cap, np.float32)
        classes = np.empty(cap, np.int32)
        index = np.empty(cap, np.int32)
        all_boxes = np.empty((n_all, 4), np.float32) if debug else None
        all_scores = np.empty((n_all, num_classes), np.float32) if debug else None
        count = C.c_int(0)
        self._check(self._lib.whenet_yolo_eval(self._h, ptrs, _ptr(gh), _ptr(gw), L, _ptr(anchors), anchors.shape[0],
                                               num_classes, float(image_shape[0]), float(image_shape[1]),
                                               float(score_threshold), float(iou_threshold), int(max_boxes), _ptr(boxes),
                                               _ptr(scores), _ptr(classes), _ptr(index), C.byref(count), _ptr(all_boxes),
                                               _ptr(all_scores)))
        k = count.value
        res = (boxes[:k].copy(), scores[:k].copy(), classes[:k].copy())
        return res + (index[:k].copy(), all_boxes, all_scores) if debug else res

    def collect(self, ticket: int, n: int, want_logits: bool = False):
        ypr = np.empty((n, 3), np.float32)
        am = np.empty((n, 3), np.int32)
        lg = np.empty((n, 252), np.float32) if want_logits else None
        self._check(self._lib.whenet_collect(self._h, ticket, _ptr(ypr), _ptr(am), _ptr(lg)))
        return ypr, am, lg

    # ---- misc -------------------------------------------------------------------------
    def set_option(self, key: str, value: int):
        self._check(self._lib.whenet_set_option(self._h, key.encode(), int(value)))

    def info(self) -> Info:
        out = Info()
        self._check(self._lib.whenet_get_info(self._h, C.byref(out)))
        return out

    def profile(self, d_crops: int, n: int, iters: int = 10):
        cap = 128
        arr = (LaunchStat * cap)()
        cnt = C.c_int(0)
        self._check(self._lib.whenet_profile(self._h, d_crops, n, iters, arr, cap, C.byref(cnt)))
        out = []
        for i in range(min(cnt.value, cap)):
            s = arr[i]
            out.append({"layer": s.layer.decode(), "kind": s.kind.decode(), "kernel": s.kernel.decode(),
                        "avg_us": s.avg_us, "alg_bytes": s.alg_bytes, "alg_flops": s.alg_flops,
                        "crops": int(s.crops), "chains": int(s.chains)})
        return out

    def device_alloc(self, nbytes: int) -> int:
        p = _P()
        self._check(self._lib.whenet_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr: int):
        self._check(self._lib.whenet_device_free(self._h, ptr))

    def h2d(self, d_ptr: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        self._check(self._lib.whenet_memcpy_h2d(self._h, d_ptr, _ptr(a), a.nbytes))

    def d2h(self, a: np.ndarray, d_ptr: int):
        self._check(self._lib.whenet_memcpy_d2h(self._h, _ptr(a), d_ptr, a.nbytes))

    # ---- single-stage entry points (tests) ----------------------------------------------
    def op_stem(self, crops: np.ndarray) -> np.ndarray:
        n = crops.shape[0]
        out = np.empty((n, 112, 112, 32), np.float32)
        self._check(self._lib.whenet_op_stem(self._h, _ptr(crops), n, _ptr(out)))
        return out

    def op_block(self, index: int, x: np.ndarray):
        from . import spec
        b = spec.blocks()[index - 1]
        x = np.ascontiguousarray(x, np.float32)
        n = x.shape[0]
        assert x.shape[1:] == (b.h_in, b.h_in, b.cin), (x.shape, b)
        ex = np.empty((n, b.h_in, b.h_in, b.cexp), np.float32) if b.has_expand else None
        dw = np.empty((n, b.h_out, b.h_out, b.cexp), np.float32)
        gate = np.empty((n, b.cexp), np.float32)
        out = np.empty((n, b.h_out, b.h_out, b.cout), np.float32)
        self._check(self._lib.whenet_op_block(self._h, index, _ptr(x), n, _ptr(ex), _ptr(dw), _ptr(gate), _ptr(out)))
        return {"expand": ex, "dw": dw, "gate": gate, "out": out}

    def op_block_range(self, first: int, last: int, x: np.ndarray) -> np.ndarray:
        """Blocks first..last as the forward pass chains them (with option fold12 when the range holds 1 and 2)."""
        from . import spec
        bi, bo = spec.blocks()[first - 1], spec.blocks()[last - 1]
        x = np.ascontiguousarray(x, np.float32)
        n = x.shape[0]
        assert x.shape[1:] == (bi.h_in, bi.h_in, bi.cin), (x.shape, bi)
        out = np.empty((n, bo.h_out, bo.h_out, bo.cout), np.float32)
        self._check(self._lib.whenet_op_block_range(self._h, first, last, _ptr(x), n, _ptr(out)))
        return out

    def op_head(self, x: np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
        n = x.shape[0]
        assert x.shape[1:] == (7, 7, 320)
        feat = np.empty((n, 1280), np.float32)
        lg = np.empty((n, 252), np.float32)
        ypr = np.empty((n, 3), np.float32)
        am = np.empty((n, 3), np.int32)
        self._check(self._lib.whenet_op_head(self._h, _ptr(x), n, _ptr(feat), _ptr(lg), _ptr(ypr), _ptr(am)))
        return {"feat": feat, "logits": lg, "ypr": ypr, "argmax": am}

    def op_decode(self, logits: np.ndarray):
        lg = np.ascontiguousarray(logits, np.float32)
        n = lg.shape[0]
        ypr = np.empty((n, 3), np.float32)
        am = np.empty((n, 3), np.int32)
        self._check(self._lib.whenet_op_decode(self._h, _ptr(lg), n, _ptr(ypr), _ptr(am)))
        return ypr, am


def block_spec(index: int):
    lib = load()
    out = (C.c_int32 * 8)()
    rc = lib.whenet_block_spec(index, C.byref(out))
    if rc != OK:
        raise_for(rc, f"whenet_block_spec({index})")
    return tuple(out)


def dw_plan(dtype: int, index: int) -> dict:
    lib = load()
    out = (C.c_int32 * 12)()
    rc = lib.whenet_dw_plan(dtype, index, C.byref(out))
    if rc != OK:
        raise_for(rc, f"whenet_dw_plan({dtype},{index})")
    keys = ("threads", "CV", "TH", "NSX", "tiles_x", "tiles_y", "chunks", "IH", "IW", "lds_bytes", "pad", "C")
    return dict(zip(keys, out))


def front_plan(dtype: int, index: int) -> dict:
    lib = load()
    out = (C.c_int32 * 12)()
    rc = lib.whenet_front_plan(dtype, index, C.byref(out))
    if rc != OK:
        raise_for(rc, f"whenet_front_plan({dtype},{index})")
    keys = ("threads", "CC", "TH", "NSX", "tiles_x", "tiles_y", "chunks", "EH", "EW", "lds_bytes", "w_off", "C")
    return dict(zip(keys, out))
