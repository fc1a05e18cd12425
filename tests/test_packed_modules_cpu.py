"""CPU: the wiring shim parses a packed module's state_dict (tensors only) and the host path of quantize_pack reproduces
the reference Quantizer's integer activations on the captured module runs."""
import numpy as np
import torch

from quantize_amd.packed import PackedConv2d, PackedLinear


def _state(g, key):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    qmin, qmax = [float(v) for v in g.get(key, "a_qmin_qmax")]
    return {"weight": t(g.get(key, "weight_packed")), "w_des": t(g.get(key, "w_des")), "w_scale": t(g.get(key, "w_scale")),
            "w_zero": t(g.get(key, "w_zero_py")), "bias": t(g.get(key, "bias")),
            "a_quantizer.scale": t(g.get(key, "a_scale")), "a_quantizer.zero": t(g.get(key, "a_zero_py")),
            "a_quantizer.qmin": torch.tensor(qmin), "a_quantizer.qmax": torch.tensor(qmax)}


def test_conv_state_and_host_quantizer(g4):
    import quantize_amd.engine as engine
    for key in g4.index:
        m = PackedConv2d.from_state_dict(_state(g4, key), stride=2, padding=(1, 1))
        a_bits, a_sign = [int(v) for v in g4.get(key, "a_bits_sign")]
        assert (m.stride, m.padding, m.a_bits) == (2, 1, a_bits)
        assert torch.equal(m._neg_w_zero, -m.w_zero) and m.weight.dtype == torch.uint8 and m.weight.dim() == 1
        x = torch.from_numpy(g4.get(key, "x"))
        xq, x_des = m.quantize(x)               # CPU tensors: the module's arithmetic + the host packer
        assert np.array_equal(engine.tunpack(xq, x_des).numpy().astype(np.float32), g4.get(key, "qx")), key
        # fake-quant route input equals the reference's dequantised activations
        ref = (g4.get(key, "qx") + g4.get(key, "a_zero_py").reshape(1, -1, 1, 1)) * g4.get(key, "a_scale").reshape(1, -1, 1, 1)
        assert np.allclose(m.fake_quant(x).numpy(), ref, rtol=0, atol=1e-6)


def test_linear_state(g6):
    for key in g6.index:
        m = PackedLinear.from_state_dict(_state(g6, key))
        assert m.a_scale.numel() == 1 and m.w_scale.numel() >= 1
