"""The oracle's OWN statement of the network geometry.  TEST INFRASTRUCTURE.

Independent of the product package on purpose: headposeestimation-whenet_amd/whenet_hip/spec.py
and csrc/spec.h carry the product's block table; a wrong pad rule or block string there must not
be able to leak into the checker.  tests/test_spec.py compares the three tables.

What is restated (un-vendored ``efficientnet==0.0.4``, instantiated by
/root/reference/whenet.py:8 as ``efn.EfficientNetB0(include_top=False, input_shape=(224,224,3))``):

* ``BLOCK_STRINGS``  efficientnet/params.py, the B0 block arguments in the package's own string
  form ``r<repeats>_k<kernel>_s<stride><stride>_e<expand>_i<in>_o<out>_se<ratio>``; decoded by
  ``decode_block_string`` the way params.py's BlockDecoder does;
* ``BN_EPSILON``     GlobalParams.batch_norm_epsilon = 1e-3;
* ``tf_same``        TensorFlow 'SAME' padding: out = ceil(in/s), total = max((out-1)*s + k - in, 0),
  before = total // 2, after = total - before (asymmetric: the extra pixel goes bottom/right);
* the repeat rule of model.py: the first block of a stage uses the stage's stride and input
  filters, the remaining repeats use stride 1 and input filters = output filters;
* SEBlock width ``max(1, int(input_filters * se_ratio))`` on the block's INPUT filters;
* identity skip iff all strides are 1 and input_filters == output_filters.
* /root/reference/whenet.py:8 (224x224x3 input), :11-13 (Dense 120 / 66 / 66), :23-24 (mean, std).
"""
from __future__ import annotations

import math
import re
from typing import List, NamedTuple, Tuple

INPUT_SIZE = 224                               # whenet.py:8
BN_EPSILON = 1e-3
STEM_FILTERS = 32
HEAD_FILTERS = 1280
BINS = (("yaw", 120), ("pitch", 66), ("roll", 66))          # whenet.py:11-13
IMAGENET_MEAN = [0.485, 0.456, 0.406]                       # whenet.py:23
IMAGENET_STD = [0.229, 0.224, 0.225]                        # whenet.py:24

BLOCK_STRINGS = (
    "r1_k3_s11_e1_i32_o16_se0.25",
    "r2_k3_s22_e6_i16_o24_se0.25",
    "r2_k5_s22_e6_i24_o40_se0.25",
    "r3_k3_s22_e6_i40_o80_se0.25",
    "r3_k5_s11_e6_i80_o112_se0.25",
    "r4_k5_s22_e6_i112_o192_se0.25",
    "r1_k3_s11_e6_i192_o320_se0.25",
)


class StageArgs(NamedTuple):
    repeats: int
    kernel: int
    stride: int
    expand: int
    filters_in: int
    filters_out: int
    se_ratio: float


class MBConv(NamedTuple):
    number: int         # 1..16
    kernel: int
    stride: int
    expand: int
    filters_in: int
    filters_out: int
    size_in: int        # square feature map
    size_out: int
    se_ratio: float

    @property
    def filters_mid(self) -> int:
        return self.filters_in * self.expand

    @property
    def se_width(self) -> int:
        return max(1, int(self.filters_in * self.se_ratio))

    @property
    def expands(self) -> bool:
        return self.expand != 1

    @property
    def identity_skip(self) -> bool:
        return self.stride == 1 and self.filters_in == self.filters_out


def decode_block_string(s: str) -> StageArgs:
    opts = {}
    for op in s.split("_"):
        m = re.match(r"([a-z]+)([0-9.]+)$", op)
        if m is None:
            raise ValueError(f"bad block string component {op!r}")
        opts[m.group(1)] = m.group(2)
    strides = opts["s"]
    if len(strides) != 2 or strides[0] != strides[1]:
        raise ValueError("strides must be a pair of equal digits")
    return StageArgs(int(opts["r"]), int(opts["k"]), int(strides[0]), int(opts["e"]), int(opts["i"]),
                     int(opts["o"]), float(opts["se"]))


def tf_same(size: int, kernel: int, stride: int) -> Tuple[int, int, int]:
    """(output size, pad before, pad after) of TensorFlow's 'SAME' rule."""
    out = int(math.ceil(size / stride))
    total = max((out - 1) * stride + kernel - size, 0)
    before = total // 2
    return out, before, total - before


def mbconv_blocks() -> List[MBConv]:
    size = tf_same(INPUT_SIZE, 3, 2)[0]            # the stem is Conv3x3 / stride 2 'same'
    out: List[MBConv] = []
    for s in BLOCK_STRINGS:
        a = decode_block_string(s)
        for rep in range(a.repeats):
            stride = a.stride if rep == 0 else 1
            fin = a.filters_in if rep == 0 else a.filters_out
            so = tf_same(size, a.kernel, stride)[0]
            out.append(MBConv(len(out) + 1, a.kernel, stride, a.expand, fin, a.filters_out, size, so, a.se_ratio))
            size = so
    return out
