"""Operator surface of the quantized linear layer: the names and call forms of the reference's
modelzoo/modules/operator/quantlinearop.py (QuantLinearOp1 :16-37, QuantLinearOp2 :40-56, quantlinear_forward :59-80),
so a QuantLinear module can call the gfx950 engine unchanged.

Dispatch is on (input.dtype, weight.dtype), operands being plain tensors or (q, des, scale, zero) tuples:

    float32 x float32  -> torch.nn.functional.linear
    uint8   x uint8    -> engine.quantlinear              (int8 MFMA GEMM for 8-bit operands)
    float32 x uint8    -> engine.quantlinear_float_input
    anything else      -> ValueError("Unsupported input and weight types.")

Mind the conventions of the reference's linear kernels: quantlinear ADDS the zero point and indexes the activation
scale by batch row (quantlinear.cu:96,115,120); quantlinear_float_input subtracts it (quantlinear_float_input.cu:82-86).
"""
import torch
from torch.autograd import Function
from torch.nn import functional as F

from .. import engine
from .quantconv2dop import _split

_PACKED_ARGS = ("input", "input_des", "input_scale", "input_zero", "weight", "weight_des", "weight_scale", "weight_zero", "bias")
_FLOAT_ARGS = ("input", "weight", "weight_des", "weight_scale", "weight_zero", "bias")


class QuantLinearOp1(Function):
    """packed activations x packed weights -> fp32 (B, O); forward only.
    Positional arguments: input, input_des, input_scale, input_zero, weight, weight_des, weight_scale, weight_zero, bias."""

    @staticmethod
    def forward(ctx, *operands):
        assert len(operands) == len(_PACKED_ARGS), _PACKED_ARGS
        return engine.quantlinear(*operands)

    @staticmethod
    def symbolic(g, *operands):
        return g.op("QuantLinearOp1", *operands)


class QuantLinearOp2(Function):
    """fp32 activations x packed weights -> fp32 (B, O); forward only.
    Positional arguments: input, weight, weight_des, weight_scale, weight_zero, bias."""

    @staticmethod
    def forward(ctx, *operands):
        assert len(operands) == len(_FLOAT_ARGS), _FLOAT_ARGS
        return engine.quantlinear_float_input(*operands)

    @staticmethod
    def symbolic(g, *operands):
        return g.op("QuantLinearOp2", *operands)


def quantlinear_forward(input, weight, bias):
    """Forward of a QuantLinear module."""
    x, x_des, x_scale, x_zero = _split(input)
    w, w_des, w_scale, w_zero = _split(weight)
    kinds = (x.dtype, w.dtype)
    if kinds == (torch.float32, torch.float32):
        return F.linear(x, w, bias)
    if kinds == (torch.uint8, torch.uint8):
        return QuantLinearOp1.apply(x, x_des, x_scale, x_zero, w, w_des, w_scale, w_zero, bias)
    if kinds == (torch.float32, torch.uint8):
        return QuantLinearOp2.apply(x, w, w_des, w_scale, w_zero, bias)
    raise ValueError("Unsupported input and weight types.")
