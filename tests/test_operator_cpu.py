"""CPU: the Python operator surface mirrors the reference's
and refuses to compute on the host."""
import pytest
import torch
import torch.nn.functional as F


def test_engine_exports_match_reference_pybind():
    # engine/kernels/pybind.cpp:9-16
    import quantize_amd.engine as engine
    names = ["tpack", "tunpack", "linear", "quantlinear", "quantlinear_float_input", "conv2d",
             "quantconv2d", "quantconv2d_float_input"]
    assert sorted(engine.__all__) == sorted(names)
    import quant_engine  # registered as a top-level module: `from quant_engine import *` works
    for n in names:
        assert callable(getattr(quant_engine, n))
    assert quant_engine.__qe_arch__ == "gfx950"


def test_reference_engine_facade_line_works_unchanged():
    # the reference's engine/__init__.py:3 is `from quant_engine import *`
    import quantize_amd.engine  # noqa: F401
    ns = {}
    exec("from quant_engine import *", ns)
    assert "quantconv2d" in ns and "tpack" in ns


def test_fp32_fp32_falls_through_to_torch():
    from quantize_amd.operator import quantconv2d_forward
    x, w, b = torch.randn(2, 4, 9, 9), torch.randn(6, 2, 3, 3), torch.randn(6)
    y = quantconv2d_forward(x, w, b, (2, 2), (1, 1), (1, 1), 2)  # dilation & groups honoured here only
    assert torch.equal(y, F.conv2d(x, w, b, (2, 2), (1, 1), (1, 1), 2))


def test_unsupported_dtype_pair():
    from quantize_amd.operator import quantconv2d_forward, quantlinear_forward
    with pytest.raises(ValueError, match="Unsupported input and weight types."):
        quantconv2d_forward(torch.zeros(4, dtype=torch.uint8), torch.zeros(4), None, 1, 0, 1, 1)
    with pytest.raises(ValueError, match="Unsupported input and weight types."):
        quantlinear_forward(torch.zeros(4, dtype=torch.int8), torch.zeros(4, dtype=torch.uint8), None)


def test_symbolic_names():
    from quantize_amd.operator import QuantConv2dOp1, QuantConv2dOp2

    class G:
        def op(self, name, *args, **kw):
            return name, len(args), kw

    assert QuantConv2dOp1.symbolic(G(), *range(9), 2, 1) == ("QuantConv2dOp1", 9, {"stride_i": 2, "padding_i": 1})
    assert QuantConv2dOp2.symbolic(G(), *range(6), 1, 0) == ("QuantConv2dOp2", 6, {"stride_i": 1, "padding_i": 0})


def test_no_cpu_compute_path_for_the_operators():
    """conv / linear refuse host tensors exactly as the reference does (CHECK_CUDA, quantconv2d.cu:13,178); only
    tpack/tunpack dispatch on the device (tests/test_tpack_host_cpu.py), as in the reference."""
    import quantize_amd.engine as engine
    des = torch.tensor([8, 1, 1, 1, 1, 1], dtype=torch.int32)
    one = torch.ones(1)
    u8 = torch.zeros(1, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        engine.quantconv2d(u8, des, one, one, u8, des, one, one, None, 1, 0)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        engine.quantconv2d_float_input(torch.zeros(1, 1, 1, 1), u8, des, one, one, None, 1, 0)
    with pytest.raises(RuntimeError, match=r"n_bits must be in the range \(0, 8\]"):
        engine.tpack(torch.zeros(8), 9, True)


def test_product_does_not_import_the_oracle():
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for root, _, files in os.walk(os.path.join(repo, "quantize_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "qe_oracle" not in text, f
