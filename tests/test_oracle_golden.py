"""CPU: the oracle (oracle/qe_oracle.c) against the golden vectors taken from the reference.

G1 is produced by the reference's own Python packer, G3's `ref_fconv`/`exact64` by torch's
F.conv2d on dequantised tensors (the reference's packed-forward fallback), G4 by running the
reference's QuantConv2d module.  See oracle/gen_golden.py.
"""
import numpy as np
import pytest

import oracle


def test_g1_tpack_bit_exact(g1):
    assert len(g1.index) == 8 * 2 * 7
    for key in g1.index:
        x = g1.get(key, "x")
        des = g1.get(key, "des")
        packed, des_o = oracle.tpack(x, int(des[0]), bool(des[1]))
        assert np.array_equal(packed, g1.get(key, "packed")), key
        assert np.array_equal(des_o, des), key
        assert packed.dtype == np.uint8 and des_o.dtype == np.int32


def test_g1_tunpack_bit_exact(g1):
    for key in g1.index:
        des = g1.get(key, "des")
        u = oracle.tunpack(g1.get(key, "packed"), des)
        ref = g1.get(key, "unpacked")
        assert u.dtype == ref.dtype, key           # int8 iff signed (tpack.cu:452-455)
        assert u.shape == ref.shape and np.array_equal(u, ref), key


def test_tpack_dtypes_agree():
    # AT_DISPATCH_ALL_TYPES_AND(Half): integer-valued input of any dtype packs identically
    rng = np.random.RandomState(0)
    x = rng.randint(-8, 8, size=203)
    ref, _ = oracle.tpack(x.astype(np.float32), 4, True)
    for dt in (np.float16, np.float64, np.int8, np.int16, np.int32, np.int64):
        got, _ = oracle.tpack(x.astype(dt), 4, True)
        assert np.array_equal(got, ref), dt


def test_tpack_truncates_toward_zero():
    # (char)x on a non-integer float truncates toward zero (tpack.cu:50)
    a, _ = oracle.tpack(np.array([2.9, -2.9, 0.5, -0.5], np.float32), 4, True)
    b, _ = oracle.tpack(np.array([2, -2, 0, 0], np.float32), 4, True)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("x,b,sign", [([0, 4], 2, False), ([-1, 0], 3, False), ([4], 3, True), ([-5], 3, True),
                                      ([float("nan")], 8, True), ([255.5], 8, False)])
def test_tpack_out_of_range(x, b, sign):
    with pytest.raises(oracle.OracleError, match="The input tensor is out of range."):
        oracle.tpack(np.array(x, np.float32), b, sign)


@pytest.mark.parametrize("b", [0, 9, -1])
def test_tpack_bad_nbits(b):
    with pytest.raises(oracle.OracleError, match=r"n_bits must be in the range \(0, 8\]"):
        oracle.tpack(np.zeros(8, np.float32), b, True)


def test_tunpack_errors():
    with pytest.raises(oracle.OracleError, match="too short"):
        oracle.tunpack(np.zeros(4, np.uint8), np.array([8, 1], np.int32))
    with pytest.raises(oracle.OracleError, match="must be torch.uint8"):
        oracle.tunpack(np.zeros(4, np.int8), np.array([8, 1, 4], np.int32))


def _run_oracle_conv(g3, key, mode, return_f64=False):
    w = (g3.get(key, "w_packed"), g3.get(key, "w_des"), g3.get(key, "w_scale"), g3.get(key, "w_zero"))
    bias = g3.get(key, "bias")
    stride, pad = [int(v) for v in g3.get(key, "stride_pad")]
    x = g3.get(key, "x")
    if x is not None:
        return oracle.quantconv2d_float_input(x, *w, bias, stride, pad, mode=mode, return_f64=return_f64)
    return oracle.quantconv2d(g3.get(key, "x_packed"), g3.get(key, "x_des"), g3.get(key, "x_scale"),
                              g3.get(key, "x_zero"), *w, bias, stride, pad, mode=mode, return_f64=return_f64)


def test_g3_conv_oracle_vs_reference_fallback(g3):
    """The restated CUDA loop agrees with F.conv2d on dequantised tensors (quantconv2d.py:207-210)."""
    assert len(g3.index) == 56
    for key in g3.index:
        exact = g3.get(key, "exact64")
        # float64 mode of the oracle vs torch float64: independent implementations of the same sum
        o32, o64 = _run_oracle_conv(g3, key, "f64", return_f64=True)
        scale = max(1.0, float(np.abs(exact).max()))
        assert np.abs(o64 - exact).max() <= 1e-12 * scale, key
        # fp32 chains vs the reference fallback in fp32: both are fp32 evaluations of the same sum
        ref32 = g3.get(key, "ref_fconv")
        K = int(np.prod(g3.get(key, "w_des")[3:]))
        tol = 4e-7 * np.sqrt(K) * scale + 1e-6
        for mode in ("fp32", "fp32_fma"):
            got = _run_oracle_conv(g3, key, mode)
            assert got.shape == ref32.shape
            assert np.abs(got - ref32).max() <= tol * 8, (key, mode)
            assert np.abs(got - exact).max() <= tol * 8, (key, mode)


def test_g3_conv_oracle_regression(g3):
    """Bit-exact pin of the two fp32 chains (mul+add, fma) the fixtures were generated with."""
    for key in g3.index:
        assert np.array_equal(_run_oracle_conv(g3, key, "fp32"), g3.get(key, "chain32")), key
        assert np.array_equal(_run_oracle_conv(g3, key, "fp32_fma"), g3.get(key, "chain32_fma")), key


def test_g4_module_capture(g4):
    """Reference QuantConv2d calibrate -> pack -> reload -> forward vs the oracle on the captured tuples.

    The modules use (q + zero) * scale (quantizer.py:218), the kernels (q - zero) * scale
    (quantconv2d.cu:113-115): zeros are negated at this boundary."""
    assert len(g4.index) == 9
    for key in g4.index:
        qx = g4.get(key, "qx")
        a_bits, a_sign = [int(v) for v in g4.get(key, "a_bits_sign")]
        xq, x_des = oracle.tpack(qx, a_bits, bool(a_sign))
        w_des = g4.get(key, "w_des")
        assert w_des.dtype == np.int32 and len(w_des) == 6
        stride, pad = [int(v) for v in g4.get(key, "stride_pad")]
        y = oracle.quantconv2d(xq, x_des, g4.get(key, "a_scale"), -g4.get(key, "a_zero_py"),
                               g4.get(key, "weight_packed"), w_des, g4.get(key, "w_scale"),
                               -g4.get(key, "w_zero_py"), g4.get(key, "bias"), stride, pad, mode="f64")
        ref = g4.get(key, "y_packed")
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), key
        # and the float-input operator on the dequantised activations
        xf = (qx + g4.get(key, "a_zero_py").reshape(1, -1, 1, 1)) * g4.get(key, "a_scale").reshape(1, -1, 1, 1)
        y2 = oracle.quantconv2d_float_input(xf.astype(np.float32), g4.get(key, "weight_packed"), w_des,
                                            g4.get(key, "w_scale"), -g4.get(key, "w_zero_py"),
                                            g4.get(key, "bias"), stride, pad, mode="f64")
        assert np.abs(y2 - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), key


def test_conv_empty_and_degenerate():
    w, wd = oracle.tpack(np.zeros((2, 3, 3, 3), np.float32), 8, True)
    x, xd = oracle.tpack(np.zeros((1, 3, 2, 2), np.float32), 8, True)
    one, zero = np.ones(1, np.float32), np.zeros(1, np.float32)
    # 2x2 input, 3x3 kernel, no padding -> OH = 0: empty output
    y = oracle.quantconv2d(x, xd, one, zero, w, wd, one, zero, None, 1, 0)
    assert y.size == 0
    # with padding 1 every border tap is skipped, result is just the bias
    b = np.array([0.5, -1.5], np.float32)
    y = oracle.quantconv2d(x, xd, one, zero, w, wd, one, zero, b, 1, 1)
    assert y.shape == (1, 2, 2, 2) and np.array_equal(y[0, :, 0, 0], b)
