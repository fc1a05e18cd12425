"""GPU: the LDS-DMA 3x3 kernel (qe_conv_halod.hip, "sm2d") vs the oracle: two 14x14 images per workgroup, 14-row bands
of larger planes (clipped at the image top / bottom), four 7x7 images per workgroup, ragged image groups and
output-channel tiles, planes whose row range is not a multiple of 16 bytes (shifted last DMA slot), 2..16 stages,
symmetric and asymmetric operands (S_x / S_w / border-class terms), each also with the kernel disabled (QE_SM2D=0)."""
import os

import numpy as np
import pytest

from test_conv_gpu import _random_case, _run_case, _assert_conv_close, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, IC, H, W, OC, K, stride, pad
    (4, 256, 14, 14, 256, 3, 1, 1),     # the ResNet-50 layer3 shape: 2 images per workgroup, 8 stages
    (3, 64, 14, 14, 130, 3, 1, 1),      # ragged image group and output-channel tile, 2 stages
    (2, 128, 28, 28, 128, 3, 1, 1),     # two 14-row bands per image (the second clipped at the bottom)
    (1, 96, 30, 30, 72, 3, 1, 1),       # 3 stages, bands of 15 / 15 rows, half-empty output-channel tile
    (2, 64, 56, 56, 128, 3, 1, 1),      # 7-row bands, 32 slots per channel row
    (5, 512, 7, 7, 512, 3, 1, 1),       # 7x7: 49-byte planes (byte-aligned sources), 16 stages
    (9, 160, 7, 7, 200, 3, 1, 1),
    (2, 64, 20, 12, 96, 3, 1, 1),       # 240-pixel planes: whole images, rows of 12 bytes
    (3, 64, 9, 8, 80, 3, 1, 1),         # narrowest rows the kernel takes (W = 8)
    (2, 96, 17, 23, 100, 3, 1, 1),      # odd sizes: bands whose byte range is not a multiple of 16
    (20, 128, 14, 14, 128, 3, 1, 1),    # more pixel tiles than XCDs
]


@pytest.mark.parametrize("sm2d", ["1", "2", "0"])
def test_sm2d_vs_oracle(engine, sm2d):
    rng = np.random.RandomState(4242)
    old = os.environ.get("QE_SM2D")
    os.environ["QE_SM2D"] = sm2d
    try:
        for shp in SHAPES:
            for (asgn, zeros) in [(1, False), (0, True), (1, True)]:
                case = _random_case(rng, *shp, 8, 1 if asgn else 0, 8, asgn, w_pc=True, a_pc=False, zeros=zeros, bias=True)
                y, o32, o64 = _run_case(engine, case, via_capi=True)
                assert case["path"] == 1
                _assert_conv_close(y, o64, o32, "sm2d=%s %s asgn=%d zeros=%s" % (sm2d, shp, asgn, zeros), case["fma"])
                if not zeros:
                    assert np.abs(y.astype(np.float64) - o64).max() <= 1e-5
    finally:
        if old is None:
            os.environ.pop("QE_SM2D", None)
        else:
            os.environ["QE_SM2D"] = old
