"""Operator surface of the quantized conv2d, mirroring the reference's
modelzoo/modules/operator/quantconv2dop.py so that QuantConv2d modules and the
runners call the gfx950 engine unchanged.

Kept identical to the reference on purpose:
  * `quantconv2d_forward(input, weight, bias, stride, padding, dilation, groups)`
    (quantconv2dop.py:69-97): `input` / `weight` may be tensors or
    `(q, des, scale, zero)` tuples; dispatch on (input.dtype, weight.dtype);
    fp32 x fp32 falls through to F.conv2d with dilation and groups (:78-79);
    tuple/list stride and padding are reduced to their first element (:82-85), so
    the quantized paths are square-stride, no dilation, no groups;
    any other dtype pair raises ValueError("Unsupported input and weight types.") (:97).
  * `QuantConv2dOp1` (packed x packed -> engine.quantconv2d, :16-41) and
    `QuantConv2dOp2` (fp32 x packed -> engine.quantconv2d_float_input, :44-66):
    forward-only autograd Functions with ONNX symbolics named after the class and
    integer attributes stride_i / padding_i.
"""
import torch
from torch.autograd import Function
from torch.nn import functional as F

from ..engine import quantconv2d, quantconv2d_float_input


def _first(v):
    """stride/padding arrive as int or (h, w); the engine is square (quantconv2dop.py:82-85)."""
    return v[0] if isinstance(v, (tuple, list)) else v


def _split(t):
    """A packed operand travels as (q, des, scale, zero) (quantconv2dop.py:72-75)."""
    if isinstance(t, (tuple, list)):
        q, des, scale, zero = t[:4]
        return q, des, scale, zero
    return t, None, None, None


class QuantConv2dOp1(Function):
    """Packed-int activations x packed-int weights (reference quantconv2dop.py:16-41).

    input / weight: 1-D torch.uint8 bit streams; *_des: int32 [n_bits, sign, *shape];
    *_scale / *_zero: fp32, one element (per tensor) or per channel; bias: fp32 or None;
    stride / padding: Python ints. Returns fp32 (N, OC, OH, OW). No backward.
    """

    @staticmethod
    def forward(ctx, input, input_des, input_scale, input_zero,
                weight, weight_des, weight_scale, weight_zero, bias, stride, padding):
        return quantconv2d(input, input_des, input_scale, input_zero,
                           weight, weight_des, weight_scale, weight_zero,
                           bias, stride, padding)

    @staticmethod
    def symbolic(g, input, input_des, input_scale, input_zero,
                 weight, weight_des, weight_scale, weight_zero, bias, stride, padding):
        return g.op("QuantConv2dOp1", input, input_des, input_scale, input_zero,
                    weight, weight_des, weight_scale, weight_zero, bias,
                    stride_i=stride, padding_i=padding)


class QuantConv2dOp2(Function):
    """fp32 activations x packed-int weights (reference quantconv2dop.py:44-66)."""

    @staticmethod
    def forward(ctx, input, weight, weight_des, weight_scale, weight_zero, bias, stride, padding):
        return quantconv2d_float_input(input, weight, weight_des, weight_scale, weight_zero,
                                       bias, stride, padding)

    @staticmethod
    def symbolic(g, input, weight, weight_des, weight_scale, weight_zero, bias, stride, padding):
        return g.op("QuantConv2dOp2", input, weight, weight_des, weight_scale, weight_zero, bias,
                    stride_i=stride, padding_i=padding)


def quantconv2d_forward(input, weight, bias, stride, padding, dilation, groups):
    """Forward of a QuantConv2d module (reference quantconv2dop.py:69-97)."""
    input, input_des, input_scale, input_zero = _split(input)
    weight, weight_des, weight_scale, weight_zero = _split(weight)

    if input.dtype == torch.float32 and weight.dtype == torch.float32:
        return F.conv2d(input, weight, bias, stride, padding, dilation, groups)

    stride, padding = _first(stride), _first(padding)

    if input.dtype == torch.uint8 and weight.dtype == torch.uint8:
        return QuantConv2dOp1.apply(input, input_des, input_scale, input_zero,
                                    weight, weight_des, weight_scale, weight_zero,
                                    bias, stride, padding)
    if input.dtype == torch.float32 and weight.dtype == torch.uint8:
        return QuantConv2dOp2.apply(input, weight, weight_des, weight_scale, weight_zero,
                                    bias, stride, padding)
    raise ValueError("Unsupported input and weight types.")
