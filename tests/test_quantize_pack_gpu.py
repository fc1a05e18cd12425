"""GPU: fused Quantizer + tpack (qe_quantize_pack) is bit-exact against
oracle.tpack(round(x / scale - zero).clamp(qmin, qmax)) -- the reference's Quantizer arithmetic (quantizer.py:31,
:215: fp32 division, subtraction, round-half-even, clamp) followed by the reference packer's layout -- per tensor and
per channel, for every bit width, and at the headline activation size through a round trip."""
import numpy as np
import pytest
import torch

import oracle
from quantize_amd import capi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, scale, zero, qmin, qmax, inner):
    x = x.astype(np.float32)
    if scale.size == 1:
        q = np.rint(x / scale[0] - zero[0])
    else:
        ch = (np.arange(x.size) // inner) % scale.size
        q = np.rint(x / scale[ch] - zero[ch])
    return np.clip(q.astype(np.float32), np.float32(qmin), np.float32(qmax))


@pytest.mark.parametrize("bits,sign", [(8, True), (8, False), (4, True), (4, False), (6, True), (3, False), (1, False), (2, True)])
def test_quantize_pack_vs_oracle(bits, sign):
    rng = np.random.RandomState(100 + bits)
    qmin, qmax = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if sign else (0, (1 << bits) - 1)
    for (shape, per_channel) in [((7,), False), ((3, 5, 7, 7), True), ((2, 16, 14, 14), True), ((2, 16, 14, 14), False),
                                 ((1, 3, 33, 31), True), ((4, 64, 7, 7), True), ((40000,), False), ((2, 6, 8193), True)]:
        x = (rng.normal(0, 1, size=shape) * (qmax - qmin) * 0.3).astype(np.float32)
        x.flat[::97] = np.float32(0.5) * np.arange(x.flat[::97].size)           # exact .5 ties: round half to even
        C = shape[1] if per_channel else 1
        scale = rng.uniform(0.4, 1.7, size=C).astype(np.float32)
        zero = rng.uniform(-3, 3, size=C).astype(np.float32) if not sign else rng.uniform(-1, 1, size=C).astype(np.float32)
        inner = int(np.prod(shape[2:])) if per_channel else 1
        q = _ref(x.reshape(-1), scale, zero, qmin, qmax, inner)
        want, _ = oracle.tpack(q, bits, sign)
        xt = torch.from_numpy(x).to(DEV)
        got, status = capi.quantize_pack(xt.reshape(-1), torch.from_numpy(scale).to(DEV), torch.from_numpy(zero).to(DEV),
                                         qmin, qmax, bits, sign, inner=inner)
        assert int(status.item()) == 0
        assert np.array_equal(got.cpu().numpy(), want), (bits, sign, shape, per_channel)


def test_module_export_matches_tpack_of_quantizer(tmp_path):
    import quantize_amd.engine  # noqa: F401
    import quant_engine
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(4, 32, 14, 14, generator=g, device=DEV) * 40
    for (scale, zero, cd) in [(torch.tensor([0.37], device=DEV), torch.tensor([-1.25], device=DEV), 1),
                              (torch.rand(32, generator=g, device=DEV) + 0.2, torch.randn(32, generator=g, device=DEV), 1)]:
        packed, des = quant_engine.quantize_pack(x, scale, zero, -128.0, 127.0, 8, True, cd)
        view = scale.view(1, -1, 1, 1) if scale.numel() > 1 else scale
        zview = zero.view(1, -1, 1, 1) if zero.numel() > 1 else zero
        q = (x / view - zview).round().clamp(-128, 127)            # Quantizer.round + clamp (quantizer.py:31,215)
        p2, d2 = quant_engine.tpack(q.contiguous(), 8, True)
        assert torch.equal(packed, p2) and torch.equal(des, d2)
    with pytest.raises(RuntimeError, match="The input tensor is out of range."):
        quant_engine.quantize_pack(x, torch.tensor([0.01], device=DEV), torch.tensor([0.0], device=DEV), -500.0, 500.0, 8, True, 1)
    with pytest.raises(RuntimeError, match="The input tensor is out of range."):
        quant_engine.quantize_pack(torch.full((8,), float("nan"), device=DEV), torch.ones(1, device=DEV), torch.zeros(1, device=DEV),
                                   -128.0, 127.0, 8, True, 0)
    # host tensors take the module's own arithmetic + the host packer
    xc = x.cpu()
    pc, dc = quant_engine.quantize_pack(xc, torch.tensor([0.37]), torch.tensor([-1.25]), -128.0, 127.0, 8, True, 1)
    pg, _ = quant_engine.quantize_pack(x, torch.tensor([0.37], device=DEV), torch.tensor([-1.25], device=DEV), -128.0, 127.0, 8, True, 1)
    assert torch.equal(pc, pg.cpu())


def test_full_size_round_trip():
    """The largest ResNet-50 activation (256, 256, 56, 56) = 205.5 M elements: pack -> unpack equals the Quantizer output."""
    g = torch.Generator(device=DEV).manual_seed(9)
    n = 256 * 256 * 56 * 56
    x = torch.empty(n, device=DEV).normal_(0, 30, generator=g)
    scale, zero = torch.tensor([0.61], device=DEV), torch.tensor([1.5], device=DEV)
    for bits in (8, 4):
        qmin, qmax = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
        packed, st = capi.quantize_pack(x, scale, zero, qmin, qmax, bits, True)
        assert int(st.item()) == 0
        u = capi.tunpack(packed, n, bits, True)
        q = (x / scale - zero).round().clamp(qmin, qmax).to(torch.int8)
        assert torch.equal(u, q)
        del u, q, packed
