"""Mirror of modelzoo/modules/operator/__init__.py:1-2."""
from .quantconv2dop import QuantConv2dOp1, QuantConv2dOp2, quantconv2d_forward
from .quantlinearop import QuantLinearOp1, QuantLinearOp2, quantlinear_forward

__all__ = ["QuantConv2dOp1", "QuantConv2dOp2", "quantconv2d_forward",
           "QuantLinearOp1", "QuantLinearOp2", "quantlinear_forward"]
