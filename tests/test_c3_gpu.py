"""GPU: the whole-image 3x3 kernel (qe_conv_c3.hip: every input channel of a 14x14 plane resident in LDS, one strip per
wave, weights streamed from L2) vs the oracle: 256 input channels, 256 and 512 output channels (one and two channel tiles),
several images, symmetric and asymmetric activations (border-aware S_w tables) and weights (S_x), no bias -- each also with
the kernel disabled (QE_C3=0: the two-strip sm2 kernel), and bit-identical between the two."""
import os

import numpy as np
import pytest

from quantize_amd import capi
from test_conv_gpu import _random_case, _run_case, _assert_conv_close, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, IC, H, W, OC, K, stride, pad
    (2, 256, 14, 14, 256, 3, 1, 1),     # ResNet-50 layer3 conv2
    (3, 256, 14, 14, 512, 3, 1, 1),     # two 256-channel tiles
    (9, 256, 14, 14, 256, 3, 1, 1),     # more images than XCDs
    (1, 256, 14, 15, 256, 3, 1, 1),     # 210-pixel planes, rows of 15 (last quad shifted by 1 pixel)
]


def _run(env, rng_seed):
    rng = np.random.RandomState(rng_seed)
    outs = []
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for shp in SHAPES:
            for (asgn, zeros, w_pc, bias) in [(1, False, True, True), (0, True, True, True), (1, True, False, False), (0, False, True, True)]:
                case = _random_case(rng, *shp, 8, 1 if asgn else 0, 8, asgn, w_pc=w_pc, a_pc=False, zeros=zeros, bias=bias)
                y, o32, o64 = _run_case(engine_mod(), case, via_capi=True)
                assert case["path"] == 1
                _assert_conv_close(y, o64, o32, "%s %s asgn=%d zeros=%s w_pc=%s" % (env, shp, asgn, zeros, w_pc), case["fma"])
                if not zeros and asgn:
                    assert np.abs(y.astype(np.float64) - o64).max() <= 1e-5
                outs.append(y)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return outs


def engine_mod():
    import quantize_amd.engine as e
    return e


def test_c3_vs_oracle_and_sm2(engine):
    a = _run({"QE_C3": "1"}, 33)
    b = _run({"QE_C3": "0"}, 33)
    # same integer sums, same epilogue arithmetic: the two kernels agree bit for bit
    for ya, yb in zip(a, b):
        assert np.array_equal(ya, yb)
