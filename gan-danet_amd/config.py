"""Run-time switches of the product path."""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass


@dataclass
class _Config:
    # MFMA operand type of the GEMM-shaped kernels: "bf16" (v_mfma_f32_32x32x16_bf16, fused flash PAM),
    # "fp16" (BASELINE config 5: the fused PAM kernels take IEEE fp16 operands, v_mfma_f32_32x32x16_f16; every other
    # kernel runs as in "bf16") or "fp32" (exact v_mfma_f32_32x32x2_f32, unfused PAM; the tight-parity mode).
    precision: str = "bf16"


config = _Config()


def set_precision(p: str) -> None:
    if p not in ("bf16", "fp16", "fp32"):
        raise ValueError("precision must be 'bf16', 'fp16' or 'fp32'")
    config.precision = p


@contextmanager
def precision(p: str):
    old = config.precision
    set_precision(p)
    try:
        yield
    finally:
        config.precision = old
