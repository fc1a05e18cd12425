"""Operator surface of the quantized linear layer, mirroring the reference's
modelzoo/modules/operator/quantlinearop.py:16-80 (QuantLinearOp1/2, quantlinear_forward).

The dispatch table is the reference's; packed x packed goes to engine.quantlinear (int8 MFMA GEMM for
8-bit operands), fp32 x packed to engine.quantlinear_float_input, fp32 x fp32 to F.linear.  Mind the
conventions of the reference's linear kernel: quantlinear ADDS the zero point and indexes the activation
scale by batch row (quantlinear.cu:96,115,120); quantlinear_float_input subtracts it (:82-86).
"""
import torch
from torch.autograd import Function
from torch.nn import functional as F

from ..engine import quantlinear, quantlinear_float_input
from .quantconv2dop import _split


class QuantLinearOp1(Function):
    """Packed x packed linear (reference quantlinearop.py:16-37)."""

    @staticmethod
    def forward(ctx, input, input_des, input_scale, input_zero,
                weight, weight_des, weight_scale, weight_zero, bias):
        return quantlinear(input, input_des, input_scale, input_zero,
                           weight, weight_des, weight_scale, weight_zero, bias)

    @staticmethod
    def symbolic(g, input, input_des, input_scale, input_zero,
                 weight, weight_des, weight_scale, weight_zero, bias):
        return g.op("QuantLinearOp1", input, input_des, input_scale, input_zero,
                    weight, weight_des, weight_scale, weight_zero, bias)


class QuantLinearOp2(Function):
    """fp32 x packed linear (reference quantlinearop.py:40-56)."""

    @staticmethod
    def forward(ctx, input, weight, weight_des, weight_scale, weight_zero, bias):
        return quantlinear_float_input(input, weight, weight_des, weight_scale, weight_zero, bias)

    @staticmethod
    def symbolic(g, input, weight, weight_des, weight_scale, weight_zero, bias):
        return g.op("QuantLinearOp2", input, weight, weight_des, weight_scale, weight_zero, bias)


def quantlinear_forward(input, weight, bias):
    """Forward of a QuantLinear module (reference quantlinearop.py:59-80)."""
    input, input_des, input_scale, input_zero = _split(input)
    weight, weight_des, weight_scale, weight_zero = _split(weight)

    if input.dtype == torch.float32 and weight.dtype == torch.float32:
        return F.linear(input, weight, bias)
    if input.dtype == torch.uint8 and weight.dtype == torch.uint8:
        return QuantLinearOp1.apply(input, input_des, input_scale, input_zero,
                                    weight, weight_des, weight_scale, weight_zero, bias)
    if input.dtype == torch.float32 and weight.dtype == torch.uint8:
        return QuantLinearOp2.apply(input, weight, weight_des, weight_scale, weight_zero, bias)
    raise ValueError("Unsupported input and weight types.")
