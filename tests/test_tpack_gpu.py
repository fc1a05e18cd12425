"""GPU: tpack/tunpack HIP kernels vs the golden vectors of the reference packer and vs the oracle.
Bit-exact is the bar (integer/byte work)."""
import numpy as np
import pytest
import torch

import oracle
from quantize_amd import capi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def engine():
    import quantize_amd.engine as e
    return e


def test_g1_golden_through_torch_module(engine, g1):
    for key in g1.index:
        x = torch.from_numpy(g1.get(key, "x")).to(DEV)
        des = g1.get(key, "des")
        packed, des_t = engine.tpack(x, int(des[0]), bool(des[1]))
        assert packed.dtype == torch.uint8 and packed.dim() == 1 and packed.device == x.device
        assert des_t.dtype == torch.int32 and des_t.device == x.device
        assert np.array_equal(packed.cpu().numpy(), g1.get(key, "packed")), key
        assert np.array_equal(des_t.cpu().numpy(), des), key
        u = engine.tunpack(packed, des_t)
        ref = g1.get(key, "unpacked")
        assert u.dtype == (torch.int8 if des[1] else torch.uint8)
        assert tuple(u.shape) == ref.shape and np.array_equal(u.cpu().numpy(), ref), key


def test_g1_golden_through_c_abi(g1):
    for key in g1.index:
        x = torch.from_numpy(g1.get(key, "x")).to(DEV)
        des = g1.get(key, "des")
        packed, status = capi.tpack(x.reshape(-1), int(des[0]), bool(des[1]))
        assert int(status.item()) == 0
        assert np.array_equal(packed.cpu().numpy(), g1.get(key, "packed")), key
        u = capi.tunpack(packed, x.numel(), int(des[0]), bool(des[1]))
        assert np.array_equal(u.cpu().numpy().reshape(-1), g1.get(key, "unpacked").reshape(-1)), key


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.float64, torch.int8, torch.uint8,
                                   torch.int16, torch.int32, torch.int64])
def test_all_input_dtypes_vs_oracle(engine, dtype):
    rng = np.random.RandomState(5)
    for b in range(1, 9):
        for sign in (False, True):
            if dtype == torch.uint8 and sign:
                continue
            lo, hi = (-(1 << (b - 1)), (1 << (b - 1)) - 1) if sign else (0, (1 << b) - 1)
            if dtype == torch.int8 and hi > 127:
                hi = 127
            for n in (1, 7, 31, 32, 33, 8191, 8192, 8193, 3 * 8192 + 5):
                xn = rng.randint(lo, hi + 1, size=n)
                x = torch.from_numpy(xn).to(dtype).to(DEV)
                packed, des = engine.tpack(x, b, sign)
                ref, ref_des = oracle.tpack(xn.astype(np.float32), b, sign)
                assert np.array_equal(packed.cpu().numpy(), ref), (dtype, b, sign, n)
                assert np.array_equal(des.cpu().numpy(), ref_des)
                u = engine.tunpack(packed, des).cpu().numpy()
                assert np.array_equal(u.astype(np.int64), xn)


def test_unaligned_views_vs_oracle():
    """Storage offsets break the 16-byte alignment the vector path wants: same bytes must come out."""
    rng = np.random.RandomState(6)
    base = torch.from_numpy(rng.randint(-64, 64, size=20000).astype(np.float32)).to(DEV)
    for off in (1, 2, 3):
        x = base[off:off + 17001]
        packed, status = capi.tpack(x, 7, True)
        ref, _ = oracle.tpack(x.cpu().numpy(), 7, True)
        assert int(status.item()) == 0 and np.array_equal(packed.cpu().numpy(), ref)
        buf = torch.zeros(ref.size + 3, dtype=torch.uint8, device=DEV)
        out, _ = capi.tpack(x, 7, True, out=buf[off:off + ref.size])   # unaligned destination
        assert np.array_equal(out.cpu().numpy(), ref)
        src = torch.zeros(ref.size + 3, dtype=torch.uint8, device=DEV)
        src[off:off + ref.size] = torch.from_numpy(ref).to(DEV)
        dst = torch.zeros(17001 + 3, dtype=torch.int8, device=DEV)
        u = capi.tunpack(src[off:off + ref.size], 17001, 7, True, out=dst[off:off + 17001])
        assert np.array_equal(u.cpu().numpy(), x.cpu().numpy().astype(np.int8))


def test_float_truncation_and_half(engine):
    x = torch.tensor([2.9, -2.9, 0.5, -0.5, 7.0, -8.0, 6.99, -7.99], device=DEV)
    p1, _ = engine.tpack(x, 4, True)
    p2, _ = engine.tpack(x.half(), 4, True)
    ref, _ = oracle.tpack(x.cpu().numpy(), 4, True)
    ref_h, _ = oracle.tpack(x.half().cpu().numpy(), 4, True)
    assert np.array_equal(p1.cpu().numpy(), ref) and np.array_equal(p2.cpu().numpy(), ref_h)


@pytest.mark.parametrize("vals,b,sign", [([0, 4], 2, False), ([-1, 0], 3, False), ([4], 3, True), ([-5], 3, True),
                                         ([float("nan")], 8, True), ([255.5], 8, False), ([128], 8, True)])
def test_out_of_range_raises(engine, vals, b, sign):
    x = torch.zeros(9000, device=DEV)
    x[8500:8500 + len(vals)] = torch.tensor(vals, device=DEV)
    with pytest.raises(RuntimeError, match="The input tensor is out of range."):
        engine.tpack(x, b, sign)
    with pytest.raises(oracle.OracleError, match="The input tensor is out of range."):
        oracle.tpack(x.cpu().numpy(), b, sign)


def test_error_messages(engine):
    x = torch.zeros(16, device=DEV)
    with pytest.raises(RuntimeError, match=r"n_bits must be in the range \(0, 8\]"):
        engine.tpack(x, 0, True)
    with pytest.raises(RuntimeError, match="x must be contiguous"):
        engine.tpack(torch.zeros(4, 4, device=DEV).t(), 8, True)
    with pytest.raises(RuntimeError, match="not implemented for 'BFloat16'"):
        engine.tpack(x.bfloat16(), 8, True)
    with pytest.raises(RuntimeError, match="numel\\(\\) == 0"):
        engine.tpack(torch.zeros(0, device=DEV), 8, True)
    p, des = engine.tpack(x, 8, True)
    with pytest.raises(RuntimeError, match="The description is too short, which should be at least 3."):
        engine.tunpack(p, des[:2])
    with pytest.raises(RuntimeError, match="The input tensor must be torch.uint8."):
        engine.tunpack(p.to(torch.int8), des)
    with pytest.raises(RuntimeError, match="shorter than its description"):
        engine.tunpack(p[:8], des)


def test_full_size_roundtrip_and_checksum(engine):
    """BASELINE sizes: the largest activation (256,256,56,56) = 205.5 M elements, and the conv1
    activations (256,3,224,224) at 4 bits.  Properties that need no CPU pass: unpack(pack(x)) == x,
    packed length, and a 64-bit checksum of the byte stream equal to the same checksum computed
    from x with torch integer ops (8-bit: byte i is x_i + 128; 4-bit: byte i is lo | hi << 4)."""
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    x = torch.randint(-128, 128, (256, 256, 56, 56), generator=g, device=DEV, dtype=torch.int8)
    p, des = engine.tpack(x, 8, True)
    assert p.numel() == x.numel() and des.tolist() == [8, 1, 256, 256, 56, 56]
    assert torch.equal(p, (x.to(torch.int16) + 128).to(torch.uint8).reshape(-1))
    assert torch.equal(engine.tunpack(p, des), x)
    del p
    xf = torch.randint(-8, 8, (256, 3, 224, 224), generator=g, device=DEV, dtype=torch.int8).float()
    p, des = engine.tpack(xf, 4, True)
    assert p.numel() == xf.numel() // 2
    codes = (xf.reshape(-1, 2) + 8).to(torch.uint8)
    assert torch.equal(p, codes[:, 0] | (codes[:, 1] << 4))
    assert torch.equal(engine.tunpack(p, des).float(), xf)
    # a ragged odd-bit stream on a large input: spot-check windows against the oracle
    x3 = torch.randint(0, 8, (40_000_003,), generator=g, device=DEV, dtype=torch.int32)
    p, des = engine.tpack(x3, 3, False)
    assert p.numel() == (x3.numel() * 3 + 7) // 8
    assert torch.equal(engine.tunpack(p, des).to(torch.int32), x3)
    n3 = x3.numel()
    for start, count in ((0, 8000), (8 * 1_000_000, 8000), (n3 - 8003, 8003)):  # starts are multiples of 8
        seg = x3[start:start + count].cpu().numpy()
        ref, _ = oracle.tpack(seg.astype(np.float32), 3, False)
        got = p[start * 3 // 8: start * 3 // 8 + ref.size].cpu().numpy()
        assert np.array_equal(got, ref), start


def test_global_avgpool_matches_torch():
    """Auxiliary entry point used by bench.py's top-1 tail."""
    from quantize_amd import capi
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    for shape in [(5, 37, 7, 7), (3, 8, 14, 14), (2, 3, 1, 1), (256, 2048, 7, 7), (1, 130, 5, 3)]:
        x = torch.randn(shape, generator=g, device=DEV)
        y = capi.global_avgpool(x)
        ref = x.double().mean(dim=(2, 3))
        assert tuple(y.shape) == shape[:2]
        assert float((y.double() - ref).abs().max()) <= 1e-6


def test_tpack_async_returns_the_flag_instead_of_blocking(engine):
    """quant_engine.tpack_async: same bytes and des as tpack, the range flag handed back as a device tensor (0 = in range)
    instead of being read back inside the call; an out-of-range input sets it where tpack raises."""
    import torch
    x = torch.randint(-8, 8, (3, 5, 7, 9), device="cuda:0").float()
    p, d = engine.tpack(x, 4, True)
    pa, da, st = engine.tpack_async(x, 4, True)
    assert torch.equal(p, pa) and torch.equal(d, da) and st.dtype == torch.int32 and int(st.item()) == 0
    x[1, 2, 3, 4] = 9.0
    with pytest.raises(RuntimeError, match="out of range"):
        engine.tpack(x, 4, True)
    _, _, st = engine.tpack_async(x, 4, True)
    assert int(st.item()) != 0
    # host tensors: the host loop checks first (and raises), the flag is always 0
    pc, dc, sc = engine.tpack_async(torch.arange(6).float(), 3, False)
    assert int(sc.item()) == 0 and pc.dtype == torch.uint8
