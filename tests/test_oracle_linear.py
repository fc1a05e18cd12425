"""CPU: the oracle's quantlinear / quantlinear_float_input restatement against the golden vectors generated
FROM the reference (oracle/gen_golden.py G5: reference packer + the reference module's packed-forward
arithmetic in fp32 and float64; G6: the reference's QuantLinear module calibrated, packed, reloaded, run)."""
import numpy as np

import oracle


def _run(g5, key, mode, return_f64=False):
    w = (g5.get(key, "w_packed"), g5.get(key, "w_des"), g5.get(key, "w_scale"), g5.get(key, "w_zero"))
    bias = g5.get(key, "bias")
    x = g5.get(key, "x")
    if x is not None:
        return oracle.quantlinear_float_input(x, *w, bias, mode=mode, return_f64=return_f64)
    return oracle.quantlinear(g5.get(key, "x_packed"), g5.get(key, "x_des"), g5.get(key, "x_scale"),
                              g5.get(key, "x_zero"), *w, bias, mode=mode, return_f64=return_f64)


def test_g5_linear_oracle_vs_reference_arithmetic(g5):
    assert len(g5.index) == 42
    for key in g5.index:
        exact = g5.get(key, "exact64")
        o32, o64 = _run(g5, key, "f64", return_f64=True)
        assert o64.shape == exact.shape
        assert np.abs(o64 - exact).max() <= 1e-12 * max(1.0, float(np.abs(exact).max())), key
        ref32 = g5.get(key, "ref_flinear")      # F.linear on the dequantised operands (quantlinear.py:158-161)
        K = int(g5.get(key, "meta")[1])
        for mode in ("fp32", "fp32_fma"):
            got = _run(g5, key, mode)
            bound = 4e-7 * K * max(1.0, float(np.abs(exact).max())) + 1e-6
            assert np.abs(got - ref32).max() <= bound, (key, mode)
            assert np.abs(got.astype(np.float64) - exact).max() <= bound, (key, mode)


def test_g5_linear_oracle_regression(g5):
    for key in g5.index:
        assert np.array_equal(_run(g5, key, "fp32"), g5.get(key, "chain32")), key
        assert np.array_equal(_run(g5, key, "fp32_fma"), g5.get(key, "chain32_fma")), key


def test_g6_linear_module_capture(g6):
    """Reference QuantLinear calibrate -> pack -> reload -> forward vs the oracle on the captured tuples.
    quantlinear takes the modules' (q + zero) convention as is; quantlinear_float_input takes (q - zero),
    so the weight zero is negated there."""
    assert len(g6.index) == 3
    for key in g6.index:
        qx = g6.get(key, "qx")
        a_bits, a_sign = [int(v) for v in g6.get(key, "a_bits_sign")]
        xq, x_des = oracle.tpack(qx, a_bits, bool(a_sign))
        w_des = g6.get(key, "w_des")
        assert w_des.dtype == np.int32 and len(w_des) == 4
        ref = g6.get(key, "y_packed")
        y = oracle.quantlinear(xq, x_des, g6.get(key, "a_scale"), g6.get(key, "a_zero_py"),
                               g6.get(key, "weight_packed"), w_des, g6.get(key, "w_scale"), g6.get(key, "w_zero_py"),
                               g6.get(key, "bias"), mode="f64")
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), key
        xf = (qx + g6.get(key, "a_zero_py").reshape(1, -1)) * g6.get(key, "a_scale").reshape(1, -1)
        y2 = oracle.quantlinear_float_input(xf.astype(np.float32), g6.get(key, "weight_packed"), w_des,
                                            g6.get(key, "w_scale"), -g6.get(key, "w_zero_py"), g6.get(key, "bias"),
                                            mode="f64")
        assert np.abs(y2 - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), key


def test_linear_degenerate():
    one, zero = np.ones(1, np.float32), np.zeros(1, np.float32)
    w, wd = np.zeros(0, np.uint8), np.array([8, 1, 3, 0], np.int32)
    x, xd = np.zeros(0, np.uint8), np.array([8, 1, 2, 0], np.int32)
    b = np.array([0.5, -1.5, 2.0], np.float32)
    y = oracle.quantlinear(x, xd, one, zero, w, wd, one, zero, b)       # K = 0: just the bias
    assert y.shape == (2, 3) and np.array_equal(y[1], b)
    w, wd = oracle.tpack(np.ones((3, 4), np.float32), 8, True)
    x, xd = oracle.tpack(np.ones((2, 5), np.float32), 8, True)
    try:
        oracle.quantlinear(x, xd, one, zero, w, wd, one, zero, None)
        assert False
    except oracle.OracleError as e:
        assert "Input and weight do not match" in str(e)
