"""GPU: the LDS-DMA ring kernel for 1x1 convolutions (qe_conv_flatd.hip) vs the oracle: all three tile variants
(224- and 160-pixel tiles of large planes, 4 x 7x7 images), planes whose rows are only 4-byte (14x14) or byte (7x7)
aligned, the patched last bytes of the tensor, ragged output-channel tiles and image groups, 2..32 stages,
symmetric and asymmetric operands (S_x / S_w terms), each also with the kernel disabled (QE_FLATD=0)."""
import os

import numpy as np
import pytest

from quantize_amd import capi
from test_conv_gpu import _random_case, _run_case, _assert_conv_close, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, IC, H, W, OC, K, stride, pad
    (2, 128, 14, 14, 256, 1, 1, 0),     # 196-byte planes: one 224-pixel tile, 4-byte aligned rows, tail patch
    (3, 256, 14, 14, 130, 1, 1, 0),     # ragged output-channel tile
    (1, 192, 14, 14, 64, 1, 1, 0),      # 3 stages, half-empty workgroup
    (2, 128, 28, 28, 128, 1, 1, 0),     # 160-pixel tiles (784 = 4.9 tiles)
    (1, 128, 56, 56, 64, 1, 1, 0),      # 14 tiles of 224 pixels
    (2, 128, 16, 16, 96, 1, 1, 0),      # 256-pixel planes: 2 tiles of 160
    (2, 128, 10, 18, 72, 1, 1, 0),      # 180-byte planes (180 % 16 == 4)
    (5, 128, 7, 7, 200, 1, 1, 0),       # 7x7: partial image group, ragged output-channel tile
    (4, 2048, 7, 7, 128, 1, 1, 0),      # 32 stages
    (9, 512, 7, 7, 256, 1, 1, 0),
    (2, 1024, 14, 14, 256, 1, 1, 0),    # a ResNet-50 layer3 shape
    (3, 256, 28, 28, 140, 1, 2, 0),     # stride 2: gathered to 14x14 first, then this kernel
    (40, 128, 14, 14, 64, 1, 1, 0),     # more pixel tiles than XCDs, half-empty output-channel tile
    (3, 128, 56, 56, 200, 1, 1, 0),     # 42 pixel tiles, ragged output-channel tile
]


@pytest.mark.parametrize("flatd", ["7", "0"])
def test_flatd_vs_oracle(engine, flatd):
    """flatd = tile-variant mask of the DMA ring kernel (7 = every variant, 0 = register-staged kernels only)."""
    rng = np.random.RandomState(2024)
    old = os.environ.get("QE_FLATD")
    os.environ["QE_FLATD"] = flatd
    try:
        for shp in SHAPES:
            for (asgn, zeros) in [(1, False), (0, True), (1, True)]:
                case = _random_case(rng, *shp, 8, 1 if asgn else 0, 8, asgn, w_pc=True, a_pc=False, zeros=zeros, bias=True)
                y, o32, o64 = _run_case(engine, case, via_capi=True)
                assert case["path"] == 1
                _assert_conv_close(y, o64, o32, "flatd=%s %s asgn=%d zeros=%s" % (flatd, shp, asgn, zeros), case["fma"])
                if not zeros:
                    assert np.abs(y.astype(np.float64) - o64).max() <= 1e-5
    finally:
        for k, v in (("QE_FLATD", old),):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
