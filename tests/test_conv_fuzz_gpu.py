"""GPU: seeded random problems across every conv kernel family (flat, small-plane flat, two-strip / warp-specialised /
halo 3x3, stem, strided-1x1 gather, sub-8-bit expansion, generic fp32) against the CPU oracle.  Shapes are biased
towards the conditions the planners branch on: planes of 49/56/64/196 pixels, IC multiples and non-multiples of 16/32,
OC around the 32/64/128 tile sizes, batch sizes that leave partial image groups."""
import numpy as np
import pytest
import torch

from test_conv_gpu import _random_case, _run_case, _assert_conv_close, engine  # noqa: F401  (fixture re-exported)

pytestmark = pytest.mark.gpu


def _shapes(rng, n):
    out = []
    hw_pool = [(7, 7), (7, 8), (8, 8), (14, 14), (12, 12), (4, 4), (5, 9), (28, 28), (16, 10), (9, 30), (56, 56), (20, 6)]
    ic_pool = [1, 3, 4, 8, 16, 24, 32, 40, 64, 96, 128, 160, 256, 300]
    oc_pool = [1, 8, 31, 32, 33, 64, 65, 96, 128, 129, 130, 200, 256]
    while len(out) < n:
        H, W = hw_pool[rng.randint(len(hw_pool))]
        IC = ic_pool[rng.randint(len(ic_pool))]
        OC = oc_pool[rng.randint(len(oc_pool))]
        K = [1, 1, 1, 3, 3, 5, 7][rng.randint(7)]
        stride = [1, 1, 1, 2, 2, 3][rng.randint(6)]
        pad = rng.randint(0, K // 2 + 2)
        N = rng.randint(1, 7)
        if (H + 2 * pad - K) // stride + 1 <= 0 or (W + 2 * pad - K) // stride + 1 <= 0:
            continue
        if N * IC * H * W * OC * K * K > 3.0e9 // 8:     # keep the oracle in seconds
            continue
        out.append((N, IC, H, W, OC, K, stride, pad))
    return out


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_fuzz_vs_oracle(engine, seed):
    rng = np.random.RandomState(seed)
    quant = [(8, 1, 8, 1), (8, 0, 8, 0), (8, 1, 8, 0), (4, 1, 4, 1), (8, 1, 4, 0), (5, 0, 7, 1), (2, 1, 8, 1), (8, 1, 0, 0)]
    paths = {0: 0, 1: 0, 2: 0}
    for k, shp in enumerate(_shapes(rng, 60)):
        wb, wsgn, ab, asgn = quant[rng.randint(len(quant))]
        case = _random_case(rng, *shp, wb, wsgn, ab, asgn, w_pc=bool(rng.randint(2)), a_pc=(ab != 0 and rng.randint(8) == 0),
                            zeros=bool(rng.randint(2)), bias=bool(rng.randint(2)))
        y, o32, o64 = _run_case(engine, case, via_capi=bool(k % 2))
        paths[case["path"]] += 1
        assert y.shape == o32.shape
        _assert_conv_close(y, o64, o32, "seed %d case %d shape %s quant %s" % (seed, k, shp, (wb, wsgn, ab, asgn)), case["fma"])
        if case["path"] == 0:
            assert np.array_equal(y, case["fma"]), "generic path not bit-exact: seed %d case %d %s" % (seed, k, shp)
    assert paths[0] > 0 and paths[1] > 20
