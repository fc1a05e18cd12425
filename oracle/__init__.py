"""CPU oracle for the `engine.kernels` hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``quantize_amd``) never does.
"""
from .qe_oracle import (  # noqa: F401
    build, tpack, tunpack, quantconv2d, quantconv2d_float_input, quantlinear, quantlinear_float_input,
    num_threads, set_num_threads, OracleError,
)
