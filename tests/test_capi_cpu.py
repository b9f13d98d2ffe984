"""C-ABI library: loads, exports every symbol include/whenet_hip.h declares, and its host-side
logic (snapshot validation, block table, depthwise tile planner, error codes) behaves --
none of this needs a GPU, and none of it computes the path."""
import os
import re

import numpy as np
import pytest

from whenet_hip import _lib, spec, weights as W

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "whenet_hip.h")).read()
    return re.findall(r"WHENET_API\s+[\w\s\*]+?\b(whenet_\w+)\s*\(", text)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(_lib.EXPORTS)      # the binding covers the whole header


def test_block_table_matches_python_spec():
    for b in spec.blocks():
        assert _lib.block_spec(b.index) == (b.k, b.s, b.expand, b.cin, b.cout, b.h_in, b.h_out, b.se_reduced)
    with pytest.raises(ValueError):
        _lib.block_spec(17)


@pytest.mark.parametrize("dtype", [_lib.F32, _lib.F16])
def test_depthwise_plans_are_valid(dtype):
    V = 8 if dtype == _lib.F16 else 4
    esz = 2 if dtype == _lib.F16 else 4
    for b in spec.blocks():
        p = _lib.dw_plan(dtype, b.index)
        assert p["C"] == b.cexp and p["pad"] == spec.same_pad(b.h_in, b.k, b.s)[1]
        cc = p["CV"] * V
        assert b.cexp % cc == 0 and p["chunks"] * cc == b.cexp
        tw = 7 * p["NSX"]
        assert p["tiles_x"] * tw == b.h_out                       # columns covered exactly
        assert p["tiles_y"] * p["TH"] >= b.h_out > (p["tiles_y"] - 1) * p["TH"]
        assert p["IH"] == (p["TH"] - 1) * b.s + b.k and p["IW"] == (tw - 1) * b.s + b.k
        lanes = (cc // 4) * p["TH"] * p["NSX"]
        assert lanes <= p["threads"] and p["threads"] in (128, 256)
        assert p["lds_bytes"] <= 64 * 1024
        assert p["lds_bytes"] >= p["IH"] * p["IW"] * cc * esz + b.k * b.k * cc * 4


@pytest.mark.parametrize("dtype", [_lib.F32, _lib.F16])
def test_depthwise_plans_keep_whole_cache_lines_and_block_1_is_the_fused_stem_tile(dtype):
    """Round 4 (host logic of csrc/dw.hip plan_dw): block 1 -- the one depthwise launch of the default forward -- keeps a pixel's
    32 channels in ONE chunk (its f32 plan used to be two 64-byte chunks of the 128-byte pixel: every cache line fetched by two
    workgroups, 222 MB for a 103 MB input), and that plan is the tile stemdw.hip is built for (16 x 14 outputs, 256 lanes) --
    otherwise the forward silently falls back to two launches.  (Sub-line chunks are penalised, not forbidden: the 5x5 stride-2
    layers of the un-fused schedule still take them when nothing else fits the LDS budget.)"""
    esz = 2 if dtype == _lib.F16 else 4
    p1 = _lib.dw_plan(dtype, 1)
    assert (p1["threads"], p1["TH"], p1["NSX"], p1["tiles_x"], p1["tiles_y"], p1["chunks"]) == (256, 16, 2, 8, 7, 1), p1
    assert p1["CV"] * 16 == 32 * esz


def test_create_error_codes(weights, tmp_path):
    # missing file -> OSError (Keras: OSError), garbage -> ValueError, truncated -> ValueError
    with pytest.raises(OSError):
        _lib.Handle(str(tmp_path / "nope.whnp"))
    with pytest.raises(ValueError):
        _lib.Handle(b"x" * 100)
    blob = W.pack(weights)
    with pytest.raises(ValueError):
        _lib.Handle(blob[: len(blob) // 2])
    bad = dict(weights)
    bad["b3/dw_bn/var"] = np.full_like(bad["b3/dw_bn/var"], -1.0)
    with pytest.raises(ValueError):
        _lib.Handle(W.pack(bad))
    bad = dict(weights)
    bad["head/conv/kernel"] = bad["head/conv/kernel"].copy()
    bad["head/conv/kernel"][0, 0, 0, 0] = np.nan
    with pytest.raises(ValueError):
        _lib.Handle(W.pack(bad))


def test_no_cpu_fallback(weights):
    """A valid snapshot on a box without a GPU must fail loudly (ENODEV), not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.WhenetError) as e:
        _lib.Handle(W.pack(weights))
    assert e.value.code == _lib.ENODEV


def test_yolo_dropin_has_no_cpu_fallback():
    """whenet_hip.yolo.yolo_eval keeps the reference's signature (yolo_v3/model.py:193-199) and, without a GPU,
    fails loudly instead of computing on the host."""
    import inspect
    import torch
    from whenet_hip import synth, yolo
    sig = inspect.signature(yolo.yolo_eval)
    assert list(sig.parameters)[:7] == ["yolo_outputs", "anchors", "num_classes", "image_shape", "max_boxes",
                                        "score_threshold", "iou_threshold"]
    assert (sig.parameters["max_boxes"].default, sig.parameters["score_threshold"].default,
            sig.parameters["iou_threshold"].default) == (20, .6, .5)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.WhenetError) as e:
        yolo.yolo_eval(synth.yolo_maps(1), synth.YOLO_ANCHORS, 1, (720, 1280))
    assert e.value.code == _lib.ENODEV


def test_dropin_module_surface():
    import whenet
    assert hasattr(whenet, "WHENet")
    assert whenet.WHENet.predict is whenet.WHENet.get_angle
    with pytest.raises(OSError):
        whenet.WHENet("definitely_missing_WHENet.h5")
    with pytest.raises(ValueError):
        whenet._as_uint8_crops(np.zeros((224, 224, 3), np.uint8))
    with pytest.raises(ValueError):
        whenet._as_uint8_crops(np.zeros((1, 224, 224, 3), np.float32) + 0.5)    # bytes path only
    ok = whenet._as_uint8_crops(np.full((2, 224, 224, 3), 7.0))
    assert ok.dtype == np.uint8 and ok.flags.c_contiguous
    assert whenet._is_byte_valued(np.full((1, 2), 7.0)) and not whenet._is_byte_valued(np.full((1, 2), 7.5))
    assert not whenet._is_byte_valued(np.full((1, 2), -1.0)) and not whenet._is_byte_valued(np.full((1, 2), 256))


def test_product_never_imports_oracle():
    """The shipped package must not reach into oracle/ (the oracle is a checker only)."""
    pkg = os.path.join(ROOT, "headposeestimation-whenet_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)
                assert "whenet_oracle" not in src and "whenet_torch" not in src, os.path.join(dp, f)


def test_header_is_valid_c(tmp_path):
    """include/whenet_hip.h is the C ABI: it must compile as C (no C++-isms), as a C caller sees it."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "use_header.c"
    src.write_text(
        '#include "whenet_hip.h"\n'
        "int probe(whenet_t* h, const uint8_t* crops, int n, float* ypr) {\n"
        "    whenet_info_t info; whenet_launch_stat_t st; int ticket = 0; int32_t rects[4];\n"
        "    const float box[4] = {1.f, 2.f, 30.f, 40.f};\n"
        "    (void)st; (void)sizeof(info);\n"
        '    if (whenet_set_option(h, "inflight", 3) != WHENET_OK) return -1;\n'
        "    if (whenet_frame_rects(720, 1280, box, 1, rects) != WHENET_OK) return -1;\n"
        "    if (whenet_submit_u8(h, crops, n, &ticket) != WHENET_OK) return -1;\n"
        "    return whenet_collect(h, ticket, ypr, 0, 0);\n"
        "}\n")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", f"-I{inc}", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_snapshot_table_overflow_is_rejected(weights):
    """offsets / sizes near 2^64 and wrapping dimension products must not pass the bounds
    check of the WHNPACK1 table (csrc/snapshot.cpp)."""
    import struct
    blob = bytearray(W.pack(weights))
    # first table entry: u16 name_len, name, u8 dtype, u8 ndim, u32 dims[ndim], u64 off, u64 nbytes
    p = 24
    (ln,) = struct.unpack_from("<H", blob, p)
    q = p + 2 + ln
    nd = blob[q + 1]
    dims_at = q + 2
    off_at = dims_at + 4 * nd
    good = bytes(blob)
    # (a) offset near 2^64: data_off + off + nb wraps around
    b = bytearray(good)
    struct.pack_into("<Q", b, off_at, (1 << 64) - 64)
    with pytest.raises(ValueError):
        _lib.Handle(bytes(b))
    # (b) dimension product wraps to the true count: dims (2^31, 2^31, 2, count/...)
    b = bytearray(good)
    struct.pack_into(f"<{nd}I", b, dims_at, *([0x80000000] * nd))
    with pytest.raises(ValueError):
        _lib.Handle(bytes(b))
    # (c) size larger than the file
    b = bytearray(good)
    struct.pack_into("<Q", b, off_at + 8, 1 << 40)
    with pytest.raises(ValueError):
        _lib.Handle(bytes(b))


def test_create_on_a_directory_is_an_os_error(tmp_path):
    with pytest.raises(OSError):
        _lib.Handle(str(tmp_path))


def test_handle_forward_rejects_anything_but_uint8_crops():
    """Handle.forward / the shard path must never hand the C side a buffer of the wrong size/dtype
    (ADVICE r1): validation happens before the handle is even touched."""
    h = _lib.Handle.__new__(_lib.Handle)          # no device needed: the checks come first
    for bad in (np.zeros((1, 224, 224, 3), np.float32), np.zeros((1, 100, 100, 3), np.uint8),
                np.zeros((2, 224, 224, 3), np.uint8)[:, ::2].repeat(2, axis=1)[:, :, ::-1], [[1, 2]]):
        with pytest.raises(ValueError):
            _lib.Handle.forward(h, bad)
    with pytest.raises(ValueError):
        _lib.Handle.forward_f32(h, np.zeros((1, 224, 224, 3), np.float64))
