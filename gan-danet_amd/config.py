"""Run-time switches of the product path."""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass, field
from typing import Dict

PRECISIONS = ("bf16", "fp16", "fp32", "mixed")

# layer classes whose operand type can be overridden one at a time (tools/parity_attribution.py: error attribution)
LAYER_CLASSES = ("dense3x3", "fuse3x3", "conv1x1", "decoder", "cam_apply", "pam", "stem", "disc", "vgg", "other")
# classes on split-bf16 ("x3") operands under "mixed": 3x3 / stride-1 convs through the pixel-major kernels on [hi | lo | hi]
# packs, everything else through the generic conv / GEMM kernels, which split their fp32 operands while staging them
# (GD_PREC_X3); the stem (and nn.Linear, HBM-bound) stays on the exact f32 MFMA
X3_CLASSES = ("dense3x3", "fuse3x3", "conv1x1", "decoder", "cam_apply", "disc", "vgg", "other")
# classes that run split-bf16 even in the 16-bit modes because it costs (almost) nothing there
FREE_X3_CLASSES = ("conv1x1", "cam_apply")


@dataclass
class _Config:
    # MFMA operand type of the GEMM-shaped kernels:
    #   "bf16"  v_mfma_f32_32x32x16_bf16 everywhere, fused flash PAM (the fastest mode); the HBM-bound 1x1-shaped products
    #           use split-bf16 operands (three bf16 MFMAs per product: still free), the stem conv the exact f32 MFMA;
    #   "fp16"  BASELINE config 5: the fused PAM kernels take IEEE fp16 operands (v_mfma_f32_32x32x16_f16), every other
    #           kernel runs as in "bf16";
    #   "fp32"  exact v_mfma_f32_32x32x2_f32 everywhere, PAM as the reference's unfused product chain (materialised
    #           N x N matrices: small tiles only);
    #   "mixed" the mode that holds the north-star 1e-3 AT the benchmark size: the fused flash PAM (fp16 operands, fp32
    #           accumulate / softmax statistics) and every other product in split-bf16 ("x3": hi*hi + lo*hi + hi*lo,
    #           2^-16 relative); exact f32 MFMA only where it is free (stem, nn.Linear, weight-space products).
    precision: str = "bf16"
    # per-layer-class override {class: "exact" | "x3" | "16"} on top of ``precision`` (attribution runs only)
    override: Dict[str, str] = field(default_factory=dict)
    # BatchNorm statistics under data parallelism: False = per-replica batch statistics (PyTorch-DDP semantics, what
    # BASELINE config 4 implies); True = SyncBN -- the statistics (forward) and the dy sums (backward) of every
    # training-mode BatchNorm2d are reduced over all ranks, so world = W on B/W samples each equals one device on B samples
    sync_bn: bool = False


config = _Config()


def _env_override() -> None:
    """GD_LAYER_OVERRIDE="stem=exact,conv1x1=exact": process-wide per-class operand override (A/B runs of bench.py)"""
    import os
    spec = os.environ.get("GD_LAYER_OVERRIDE", "")
    for item in filter(None, (t.strip() for t in spec.split(","))):
        k, _, v = item.partition("=")
        if k not in LAYER_CLASSES or v not in ("exact", "x3", "16"):
            raise ValueError(f"GD_LAYER_OVERRIDE: bad item {item!r}")
        config.override[k] = v


_env_override()


def set_precision(p: str) -> None:
    if p not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}")
    config.precision = p


@contextmanager
def precision(p: str):
    old = config.precision
    set_precision(p)
    try:
        yield
    finally:
        config.precision = old


def set_sync_bn(on: bool) -> None:
    config.sync_bn = bool(on)


@contextmanager
def layer_override(**kw: str):
    """``layer_override(dense3x3="exact")``: run one layer class exact (or 16-bit) whatever the configured mode"""
    for k, v in kw.items():
        if k not in LAYER_CLASSES or v not in ("exact", "x3", "16"):
            raise ValueError(f"layer_override: {k}={v}")
    old = dict(config.override)
    config.override.update(kw)
    try:
        yield
    finally:
        config.override = old


def operand_mode(layer: str) -> str:
    """"16" (plain 16-bit MFMA operands), "x3" (split-bf16: hi*hi + lo*hi + hi*lo, ~2^-16 relative) or "exact" (f32 MFMA)
    for a layer class right now"""
    o = config.override.get(layer)
    if o is not None:
        return o
    if layer == "pam":
        return "exact" if config.precision == "fp32" else "16"
    if layer == "stem":
        # the generator's first conv (Cin = 8: 72 products per output) is HBM-bound: exact f32 MFMA costs nothing and, alone,
        # takes the 16-bit modes' output error from 2.5e-2 to 1.5e-2 (profiles/r03_parity_attribution.json)
        return "exact"
    if config.precision in ("bf16", "fp16"):
        # the 1x1-shaped products (projections, transitions, skip convs, CAM apply) are HBM-bound: the 1x1 kernel splits its
        # fp32 operands into hi + lo bf16 while staging them for +4 % of its time (1.20 against 1.16 ms at 184 -> 184), which
        # takes their share out of the 16-bit modes' output error (1.5e-2 -> 1.2e-2 on the reference fixture)
        return "x3" if layer in FREE_X3_CLASSES else "16"
    if config.precision == "mixed" and layer in X3_CLASSES:
        return "x3"
    return "exact"


def sixteen_bit(layer: str) -> bool:
    """does this layer class run on plain 16-bit MFMA operands right now?"""
    return operand_mode(layer) == "16"
