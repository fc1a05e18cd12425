"""Mirror of the reference's `engine` facade (engine/__init__.py:1-5): re-exports
tpack, tunpack, linear, quantlinear, quantlinear_float_input, conv2d, quantconv2d,
quantconv2d_float_input from the native `quant_engine` module.

Unlike the reference there is no `except ImportError: from utils import *` branch
(which is broken there anyway): without the built extension this
import fails.
"""
from ..loader import load_quant_engine as _load

_qe = _load()

tpack = _qe.tpack
tunpack = _qe.tunpack
linear = _qe.linear
quantlinear = _qe.quantlinear
quantlinear_float_input = _qe.quantlinear_float_input
conv2d = _qe.conv2d
quantconv2d = _qe.quantconv2d
quantconv2d_float_input = _qe.quantconv2d_float_input

# extensions of this engine (not part of the reference's facade, hence not in __all__): fused Quantizer + tpack, and
# the host-side caches of the binding
quantize_pack = _qe.quantize_pack
tpack_async = _qe.tpack_async      # tpack without the blocking read-back of the range flag: [packed, des, status]
clear_cache = _qe.clear_cache
cache_stats = _qe.cache_stats

__all__ = ["tpack", "tunpack", "linear", "quantlinear", "quantlinear_float_input",
           "conv2d", "quantconv2d", "quantconv2d_float_input"]
