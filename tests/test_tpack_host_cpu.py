"""CPU: tpack/tunpack on CPU-resident tensors -- the reference's host dispatch (tpack.cu:241-251, :458-468,
tpack_cpu :140-190, tunpack_cpu :371-419) -- bit-exact against the G1 golden vectors of the reference packer
and against the oracle for every input dtype, bit width, sign and ragged length."""
import numpy as np
import pytest
import torch

import oracle


@pytest.fixture(scope="module")
def engine():
    import quantize_amd.engine as e
    return e


def test_g1_golden_on_cpu_tensors(engine, g1):
    for key in g1.index:
        x = torch.from_numpy(g1.get(key, "x"))
        des = g1.get(key, "des")
        packed, des_t = engine.tpack(x, int(des[0]), bool(des[1]))
        assert packed.dtype == torch.uint8 and packed.dim() == 1 and packed.device.type == "cpu"
        assert des_t.dtype == torch.int32 and des_t.device.type == "cpu"
        assert np.array_equal(packed.numpy(), g1.get(key, "packed")), key
        assert np.array_equal(des_t.numpy(), des), key
        u = engine.tunpack(packed, des_t)
        ref = g1.get(key, "unpacked")
        assert u.dtype == (torch.int8 if des[1] else torch.uint8)
        assert tuple(u.shape) == ref.shape and np.array_equal(u.numpy(), ref), key


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.float64, torch.int8, torch.uint8,
                                   torch.int16, torch.int32, torch.int64])
def test_all_input_dtypes_vs_oracle(engine, dtype):
    rng = np.random.RandomState(11)
    for b in range(1, 9):
        for sign in (False, True):
            if dtype == torch.uint8 and sign:
                continue
            lo, hi = (-(1 << (b - 1)), (1 << (b - 1)) - 1) if sign else (0, (1 << b) - 1)
            if dtype == torch.int8 and hi > 127:
                hi = 127
            for n in (1, 5, 7, 8, 9, 63, 64, 65, 1000, 4099):
                xn = rng.randint(lo, hi + 1, size=n)
                x = torch.from_numpy(xn).to(dtype)
                packed, des = engine.tpack(x, b, sign)
                op, od = oracle.tpack(xn, b, sign)
                assert np.array_equal(packed.numpy(), op), (dtype, b, sign, n)
                assert np.array_equal(des.numpy(), od)
                u = engine.tunpack(packed, des)
                assert np.array_equal(u.numpy().astype(np.int64), xn), (dtype, b, sign, n)


def test_reference_shapes_and_errors(engine):
    # a 16x8x3x3 W8 layer as QuantConv2d.pack() feeds it: integer-valued fp32
    w = torch.randint(-128, 128, (16, 8, 3, 3)).float()
    packed, des = engine.tpack(w, 8, True)
    assert tuple(packed.shape) == (1152,) and des.tolist() == [8, 1, 16, 8, 3, 3]
    assert torch.equal(engine.tunpack(packed, des), w.to(torch.int8))
    # n = 5, b = 3 -> 2 bytes [27, 48] (probed on the reference's CPU path)
    p, d = engine.tpack(torch.tensor([3., 3., 0., 0., 3.]), 3, False)
    assert p.tolist() == [27, 48] and d.tolist() == [3, 0, 5]
    with pytest.raises(RuntimeError, match="The input tensor is out of range."):
        engine.tpack(torch.tensor([0., 8.]), 3, False)
    with pytest.raises(RuntimeError, match="The input tensor is out of range."):
        engine.tpack(torch.tensor([-5., 0.]), 3, True)
    with pytest.raises(RuntimeError, match="The input tensor is out of range."):
        engine.tpack(torch.tensor([float("nan")]), 8, True)
    with pytest.raises(RuntimeError, match=r"n_bits must be in the range \(0, 8\]"):
        engine.tpack(torch.zeros(8), 0, True)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        engine.tpack(torch.zeros(4, 4).t(), 8, True)
    with pytest.raises(RuntimeError, match="The input tensor must be torch.uint8."):
        engine.tunpack(torch.zeros(8, dtype=torch.int8), torch.tensor([8, 1, 8], dtype=torch.int32))
    with pytest.raises(RuntimeError, match="The description is too short"):
        engine.tunpack(torch.zeros(8, dtype=torch.uint8), torch.tensor([8, 1], dtype=torch.int32))
    with pytest.raises(RuntimeError, match="shorter than its description"):
        engine.tunpack(torch.zeros(3, dtype=torch.uint8), torch.tensor([8, 1, 8], dtype=torch.int32))
