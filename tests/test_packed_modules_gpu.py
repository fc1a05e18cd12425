"""GPU: the module wiring shim (quantize_amd/packed.py) on the captured runs of the
reference's own modules (tests/golden/g4_module.npz, g6_linear_module.npz: QuantConv2d / QuantLinear calibrated, packed,
reloaded and run through their packed forward).  From the raw fp32 input and the packed state alone -- no reference code
here -- both operator routes must reproduce the reference module's output, and the on-device quantisation must reproduce
the module's integer activations bit for bit."""
import numpy as np
import pytest
import torch

from quantize_amd.packed import PackedConv2d, PackedLinear

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _state(g, key):
    """The state_dict entries pack() leaves behind, as the reference names them."""
    qmin, qmax = [float(v) for v in g.get(key, "a_qmin_qmax")]
    return {"weight": _t(g.get(key, "weight_packed")), "w_des": _t(g.get(key, "w_des")), "w_scale": _t(g.get(key, "w_scale")),
            "w_zero": _t(g.get(key, "w_zero_py")), "bias": _t(g.get(key, "bias")),
            "a_quantizer.scale": _t(g.get(key, "a_scale")), "a_quantizer.zero": _t(g.get(key, "a_zero_py")),
            "a_quantizer.qmin": torch.tensor(qmin), "a_quantizer.qmax": torch.tensor(qmax)}


def test_packed_conv2d_on_reference_captures(g4):
    import quantize_amd.engine as engine
    for key in g4.index:
        stride, pad = [int(v) for v in g4.get(key, "stride_pad")]
        m = PackedConv2d.from_state_dict(_state(g4, key), stride=(stride, stride), padding=(pad, pad))
        a_bits, a_sign = [int(v) for v in g4.get(key, "a_bits_sign")]
        assert m.a_bits == a_bits and (m.a_signed or not a_sign)
        x = _t(g4.get(key, "x"))
        ref = g4.get(key, "y_packed")
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        # the device-side Quantizer + packer reproduces the module's integer activations exactly
        xq, x_des = m.quantize(x)
        q = engine.tunpack(xq, x_des).cpu().numpy().astype(np.float32)
        assert np.array_equal(q, g4.get(key, "qx")), key
        for route in ("packed", "float"):
            y = m(x, route=route)
            assert y.dtype == torch.float32 and tuple(y.shape) == ref.shape
            assert np.abs(y.cpu().numpy() - ref).max() <= tol, (key, route)
    with pytest.raises(ValueError):
        m(x, route="other")


def test_packed_linear_on_reference_captures(g6):
    import quantize_amd.engine as engine
    for key in g6.index:
        m = PackedLinear.from_state_dict(_state(g6, key))
        x = _t(g6.get(key, "x"))
        ref = g6.get(key, "y_packed")
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        xq, x_des = m.quantize(x, channel_dim=1)
        assert np.array_equal(engine.tunpack(xq, x_des).cpu().numpy().astype(np.float32), g6.get(key, "qx")), key
        for route in ("packed", "float"):
            y = m(x, route=route)
            assert np.abs(y.cpu().numpy() - ref).max() <= tol, (key, route)
        # leading dimensions (tokens) are flattened and restored
        y3 = m(x.reshape(3, 4, -1), route="packed")
        assert tuple(y3.shape) == (3, 4, ref.shape[1]) and np.abs(y3.reshape(12, -1).cpu().numpy() - ref).max() <= tol


def _synthetic_state(rng, IC, OC, K, a_signed):
    """A packed QuantConv2d's state_dict entries (quantconv2d.py:187-192) with random codes and per-channel weight scales."""
    import quantize_amd.engine as engine
    qw = torch.from_numpy(rng.randint(-128, 128, size=(OC, IC, K, K)).astype(np.int8)).to(DEV)
    wp, wd = engine.tpack(qw, 8, True)
    qmin, qmax = (-128.0, 127.0) if a_signed else (0.0, 255.0)
    return {"weight": wp, "w_des": wd, "w_scale": _t(rng.uniform(1e-3, 3e-3, size=(OC, 1, 1, 1)).astype(np.float32)),
            "w_zero": _t(np.zeros((OC, 1, 1, 1), np.float32)), "bias": _t(rng.normal(0, 0.1, size=OC).astype(np.float32)),
            "a_quantizer.scale": _t(np.array([0.02 if a_signed else 0.011], np.float32)),
            "a_quantizer.zero": _t(np.array([0.0], np.float32)),
            "a_quantizer.qmin": torch.tensor(qmin), "a_quantizer.qmax": torch.tensor(qmax)}


def test_packed_layers_chain_without_fp32_in_between():
    """Three packed convs chained: each layer's conv kernel writes the NEXT layer's activation codes (fused
    re-quantisation) -- same final output, bit for bit, as quantising every fp32 intermediate with the consumer's
    quantizer (quantize_pack), which test_packed_conv2d_on_reference_captures pins to the reference module."""
    rng = np.random.RandomState(11)
    a = PackedConv2d.from_state_dict(_synthetic_state(rng, 64, 128, 1, True), stride=1, padding=0)
    b = PackedConv2d.from_state_dict(_synthetic_state(rng, 128, 128, 3, False), stride=1, padding=1)    # unsigned: ReLU folded
    c = PackedConv2d.from_state_dict(_synthetic_state(rng, 128, 256, 1, True), stride=2, padding=0)
    # the consumers' scales must suit what arrives: calibrate them from one fp32 pass
    x = torch.randn(4, 64, 28, 28, device=DEV)
    xq, xd = a.quantize(x)
    ya = a.call_packed(xq, xd)
    b.a_scale = (ya.clamp(min=0).max() / 255.0).reshape(1)
    yb = b.call_packed(*b.quantize(ya))
    c.a_scale = (yb.abs().max() / 127.0).reshape(1)
    ref = c.call_packed(*c.quantize(yb))
    # fused chain
    qb, db = a.call_packed(xq, xd, consumer=b)
    qb_ref, db_ref = b.quantize(ya)
    assert torch.equal(qb, qb_ref) and torch.equal(db.cpu(), db_ref.cpu())
    qc, dc = b.call_packed(qb, db, consumer=c)
    out = c.call_packed(qc, dc)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert tuple(out.shape) == (4, 256, 14, 14)
