"""EfficientNet-B0 (no top) + WHENet heads, as data.

This is the *shape contract* of the hot path.  It restates
what the reference builds at /root/reference/whenet.py:8-14 --
``efn.EfficientNetB0(include_top=False, input_shape=(224,224,3))`` from the un-vendored
pip package ``efficientnet==0.0.4`` followed by GAP + Dense(120/66/66) -- as a flat list
of layers with their Keras-native weight shapes.  The C++ engine has its own copy of the
block table (csrc/spec.h); tests/test_spec.py checks the two agree through the C-ABI.

Nothing here computes; it is shared by the weight tools, the oracle, the tests and the
host shim.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

IMG = 224                      # whenet.py:8  input_shape=(224, 224, 3)
BN_EPS = 1e-3                  # efficientnet 0.0.4 GlobalParams.batch_norm_epsilon
MEAN = (0.485, 0.456, 0.406)   # whenet.py:23
STD = (0.229, 0.224, 0.225)    # whenet.py:24
N_YAW, N_PITCH, N_ROLL = 120, 66, 66          # whenet.py:11-13
N_LOGITS = N_YAW + N_PITCH + N_ROLL            # 252
FEAT = 1280
STEM_C = 32

# efficientnet 0.0.4 params.py block strings for B0 (width 1.0 / depth 1.0):
#   r1_k3_s11_e1_i32_o16_se0.25 ... r1_k3_s11_e6_i192_o320_se0.25
# (repeats, kernel, stride, expand, in, out)
STAGES: Tuple[Tuple[int, int, int, int, int, int], ...] = (
    (1, 3, 1, 1, 32, 16),
    (2, 3, 2, 6, 16, 24),
    (2, 5, 2, 6, 24, 40),
    (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112),
    (4, 5, 2, 6, 112, 192),
    (1, 3, 1, 6, 192, 320),
)
SE_RATIO = 0.25


def same_pad(in_size: int, k: int, s: int) -> Tuple[int, int, int]:
    """TensorFlow 'SAME' padding: returns (out_size, pad_before, pad_after)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


@dataclass(frozen=True)
class Block:
    index: int          # 1..16
    k: int
    s: int
    expand: int
    cin: int
    cout: int
    h_in: int           # input spatial size (square)
    h_out: int

    @property
    def cexp(self) -> int:
        return self.cin * self.expand

    @property
    def se_reduced(self) -> int:
        # SEBlock: max(1, int(block.input_filters * se_ratio)) -- INPUT filters of the block
        return max(1, int(self.cin * SE_RATIO))

    @property
    def has_expand(self) -> bool:
        return self.expand != 1

    @property
    def has_skip(self) -> bool:
        return self.s == 1 and self.cin == self.cout


def blocks() -> List[Block]:
    out: List[Block] = []
    h = same_pad(IMG, 3, 2)[0]      # after the stride-2 stem: 112
    idx = 0
    for (r, k, s, e, cin, cout) in STAGES:
        for j in range(r):
            idx += 1
            bs = s if j == 0 else 1
            bcin = cin if j == 0 else cout
            h_out = same_pad(h, k, bs)[0]
            out.append(Block(idx, k, bs, e, bcin, cout, h, h_out))
            h = h_out
    return out


@dataclass(frozen=True)
class Tensor:
    name: str
    shape: Tuple[int, ...]


def _bn(prefix: str, c: int) -> List[Tensor]:
    # Keras BatchNormalization weight order: gamma, beta, moving_mean, moving_variance
    return [Tensor(f"{prefix}/gamma", (c,)), Tensor(f"{prefix}/beta", (c,)),
            Tensor(f"{prefix}/mean", (c,)), Tensor(f"{prefix}/var", (c,))]


def tensors() -> List[Tensor]:
    """All 315 weight arrays in Keras creation order, Keras-native layouts
    (Conv2D HWIO, DepthwiseConv2D (kh,kw,C,1), Dense (in,out)) -- SURVEY.md Appendix C."""
    t: List[Tensor] = [Tensor("stem/conv/kernel", (3, 3, 3, STEM_C))]
    t += _bn("stem/bn", STEM_C)
    for b in blocks():
        p = f"b{b.index}"
        if b.has_expand:
            t.append(Tensor(f"{p}/expand/kernel", (1, 1, b.cin, b.cexp)))
            t += _bn(f"{p}/expand_bn", b.cexp)
        t.append(Tensor(f"{p}/dw/kernel", (b.k, b.k, b.cexp, 1)))
        t += _bn(f"{p}/dw_bn", b.cexp)
        t.append(Tensor(f"{p}/se_reduce/kernel", (1, 1, b.cexp, b.se_reduced)))
        t.append(Tensor(f"{p}/se_reduce/bias", (b.se_reduced,)))
        t.append(Tensor(f"{p}/se_expand/kernel", (1, 1, b.se_reduced, b.cexp)))
        t.append(Tensor(f"{p}/se_expand/bias", (b.cexp,)))
        t.append(Tensor(f"{p}/project/kernel", (1, 1, b.cexp, b.cout)))
        t += _bn(f"{p}/project_bn", b.cout)
    t.append(Tensor("head/conv/kernel", (1, 1, 320, FEAT)))
    t += _bn("head/bn", FEAT)
    for name, n in (("yaw", N_YAW), ("pitch", N_PITCH), ("roll", N_ROLL)):
        t.append(Tensor(f"{name}/kernel", (FEAT, n)))
        t.append(Tensor(f"{name}/bias", (n,)))
    return t


def param_count() -> Tuple[int, int]:
    """(backbone, heads) parameter counts: must be (4_049_564, 322_812)."""
    bb = hd = 0
    for x in tensors():
        n = 1
        for d in x.shape:
            n *= d
        if x.name.split("/")[0] in ("yaw", "pitch", "roll"):
            hd += n
        else:
            bb += n
    return bb, hd


def bn_names() -> List[str]:
    """The 49 BatchNorm prefixes in creation order."""
    seen: List[str] = []
    for x in tensors():
        if x.name.endswith("/gamma"):
            seen.append(x.name[: -len("/gamma")])
    return seen


@dataclass
class Work:
    """Algorithmic work per crop for one layer class."""
    macs: int = 0
    in_elems: int = 0
    out_elems: int = 0
    layers: List[Tuple[str, int, int, int]] = field(default_factory=list)  # (name, macs, in, out)

    def add(self, name: str, macs: int, i: int, o: int) -> None:
        self.macs += macs
        self.in_elems += i
        self.out_elems += o
        self.layers.append((name, macs, i, o))


def work_table() -> dict:
    """MACs and activation element counts per layer class, per crop."""
    w = {k: Work() for k in ("stem", "pw", "dw", "se", "fc")}
    h = same_pad(IMG, 3, 2)[0]
    w["stem"].add("stem", h * h * 27 * STEM_C, IMG * IMG * 3, h * h * STEM_C)
    for b in blocks():
        p = f"b{b.index}"
        hi, ho = b.h_in, b.h_out
        if b.has_expand:
            w["pw"].add(f"{p}/expand", hi * hi * b.cin * b.cexp, hi * hi * b.cin, hi * hi * b.cexp)
        w["dw"].add(f"{p}/dw", ho * ho * b.k * b.k * b.cexp, hi * hi * b.cexp, ho * ho * b.cexp)
        w["se"].add(f"{p}/se", 2 * b.cexp * b.se_reduced, ho * ho * b.cexp, b.cexp)
        w["pw"].add(f"{p}/project", ho * ho * b.cexp * b.cout, ho * ho * b.cexp, ho * ho * b.cout)
    w["pw"].add("head", 49 * 320 * FEAT, 49 * 320, 49 * FEAT)
    w["fc"].add("heads", FEAT * N_LOGITS, FEAT, N_LOGITS)
    return w


TOTAL_MACS = 384_857_312          # SURVEY.md Appendix A total; asserted in tests/test_spec.py
