"""GPU: quantlinear / quantlinear_float_input (HIP, through the torch module and through the C ABI) against the
golden vectors from the reference (G5, G6) and the CPU oracle on random problems, including ViT-B/16 shapes."""
import numpy as np
import pytest
import torch

import oracle
from conftest import conv_tolerance
from quantize_amd import capi

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.fixture(scope="module")
def engine():
    import quantize_amd.engine as e
    return e


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _close(got, exact64, chain32, what, fma=None):
    err, allowed = conv_tolerance(got, exact64, chain32, fma)
    bad = err > allowed
    assert not bad.any(), "%s: %d elements off, worst err %.3g (allowed %.3g)" % (
        what, int(bad.sum()), float(err.max()), allowed)


def test_g5_golden_cases(engine, g5):
    paths = set()
    fpaths = set()
    for key in g5.index:
        B, K, O, wb, wsgn, ab, asgn = [int(v) for v in g5.get(key, "meta")]
        w = (_t(g5.get(key, "w_packed")), _t(g5.get(key, "w_des")), _t(g5.get(key, "w_scale")), _t(g5.get(key, "w_zero")))
        bias = g5.get(key, "bias")
        bias = None if bias is None else _t(bias)
        x = g5.get(key, "x")
        if x is not None:
            xt = _t(x)
            y = engine.quantlinear_float_input(xt, *w, bias)
            fpath = capi.linear_float_input_path(xt, capi.qparam(w[0], wb, wsgn, w[2], w[3]), B, K, O)
            fpaths.add(fpath)
            # the fp32 kernel keeps the reference's k-sequential fmaf chain: bit-identical to the oracle's (the bf16 x 3
            # MFMA kernel -- 8-bit weights, K % 32 == 0 -- is held to the tolerance below)
            if fpath == 0:
                assert np.array_equal(y.cpu().numpy(), g5.get(key, "chain32_fma")), key
        else:
            xs, xz = _t(g5.get(key, "x_scale")), _t(g5.get(key, "x_zero"))
            y = engine.quantlinear(_t(g5.get(key, "x_packed")), _t(g5.get(key, "x_des")), xs, xz, *w, bias)
            xq = capi.qparam(_t(g5.get(key, "x_packed")), ab, asgn, xs, xz)
            wq = capi.qparam(w[0], wb, wsgn, w[2], w[3])
            path = capi.linear_path(xq, wq, B, K, O)
            paths.add(path)
            if path == 0:
                assert np.array_equal(y.cpu().numpy(), g5.get(key, "chain32_fma")), key
        ref = g5.get(key, "exact64")
        assert y.dtype == torch.float32 and tuple(y.shape) == ref.shape and y.is_contiguous()
        # chain32 = the reference module's packed forward (fp32 F.linear); chain32_fma = the reference kernel's own
        # k-sequential chain (oracle) -- both are legitimate fp32 evaluations of the reference
        _close(y.cpu().numpy(), ref, g5.get(key, "chain32"), key, g5.get(key, "chain32_fma"))
    assert paths == {0, 1}      # both the MFMA GEMM and the fp32 kernel were exercised


def test_g6_reference_module_capture(engine, g6):
    from quantize_amd.operator import quantlinear_forward
    for key in g6.index:
        qx = _t(g6.get(key, "qx"))
        a_bits, a_sign = [int(v) for v in g6.get(key, "a_bits_sign")]
        xq, x_des = engine.tpack(qx, a_bits, bool(a_sign))
        wp, wd = _t(g6.get(key, "weight_packed")), _t(g6.get(key, "w_des"))
        ws, wz = _t(g6.get(key, "w_scale")), _t(g6.get(key, "w_zero_py"))
        bias = _t(g6.get(key, "bias"))
        ref = g6.get(key, "y_packed")
        scale = max(1.0, float(np.abs(ref).max()))
        # quantlinear takes the modules' (q + zero) convention unchanged
        y = quantlinear_forward((xq, x_des, _t(g6.get(key, "a_scale")).reshape(-1), _t(g6.get(key, "a_zero_py")).reshape(-1)),
                                (wp, wd, ws.reshape(-1).contiguous(), wz.reshape(-1).contiguous()), bias)
        assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * scale, key
        xf = (qx + _t(g6.get(key, "a_zero_py")).view(1, -1)) * _t(g6.get(key, "a_scale")).view(1, -1)
        y2 = quantlinear_forward(xf.contiguous(), (wp, wd, ws.reshape(-1).contiguous(), (-wz).reshape(-1).contiguous()), bias)
        assert np.abs(y2.cpu().numpy() - ref).max() <= 2e-5 * scale, key


def _random_case(rng, B, K, O, wb, wsgn, ab, asgn, w_pc, a_pr, zeros, bias):
    rngq = lambda b, s: (-(1 << (b - 1)), (1 << (b - 1)) - 1) if s else (0, (1 << b) - 1)
    wlo, whi = rngq(wb, wsgn)
    qw = rng.randint(wlo, whi + 1, size=(O, K))
    n_ws = O if w_pc else 1
    sw = rng.uniform(2.5e-4, 7.5e-4, size=n_ws).astype(np.float32)
    zw = rng.uniform(-3, 3, size=n_ws).astype(np.float32) if zeros else np.zeros(n_ws, np.float32)
    wp, wd = oracle.tpack(qw.astype(np.float32), wb, bool(wsgn))
    b = rng.normal(0, 0.1, size=O).astype(np.float32) if bias else None
    case = dict(B=B, K=K, O=O, wp=wp, wd=wd, sw=sw, zw=zw, bias=b, wb=wb, wsgn=wsgn, ab=ab, asgn=asgn)
    if ab == 0:
        x = rng.normal(0, 1, size=(B, K)).astype(np.float32)
        case["x"] = x
        case["o32"] = oracle.quantlinear_float_input(x, wp, wd, sw, zw, b, mode="fp32")
        case["fma"] = oracle.quantlinear_float_input(x, wp, wd, sw, zw, b, mode="fp32_fma")
        case["o64"] = oracle.quantlinear_float_input(x, wp, wd, sw, zw, b, mode="f64", return_f64=True)[1]
    else:
        alo, ahi = rngq(ab, asgn)
        qx = rng.randint(alo, ahi + 1, size=(B, K))
        n_as = B if a_pr else 1
        sx = rng.uniform(1e-3, 3e-3, size=n_as).astype(np.float32)
        zx = rng.uniform(-5, 5, size=n_as).astype(np.float32) if zeros else np.zeros(n_as, np.float32)
        xp, xd = oracle.tpack(qx.astype(np.float32), ab, bool(asgn))
        case.update(xp=xp, xd=xd, sx=sx, zx=zx)
        case["o32"] = oracle.quantlinear(xp, xd, sx, zx, wp, wd, sw, zw, b, mode="fp32")
        case["fma"] = oracle.quantlinear(xp, xd, sx, zx, wp, wd, sw, zw, b, mode="fp32_fma")
        case["o64"] = oracle.quantlinear(xp, xd, sx, zx, wp, wd, sw, zw, b, mode="f64", return_f64=True)[1]
    return case


def _run(engine, c, via_capi):
    bias = None if c["bias"] is None else _t(c["bias"])
    wp, sw, zw = _t(c["wp"]), _t(c["sw"]), _t(c["zw"])
    if "x" in c:
        xt = _t(c["x"])
        wq = capi.qparam(wp, c["wb"], c["wsgn"], sw, zw)
        fpath = 2 * capi.linear_float_input_path(xt, wq, c["B"], c["K"], c["O"])     # 0: fp32 chain kernel, 2: bf16 x 3 MFMA kernel
        if via_capi:
            return capi.quantlinear_float_input(xt, wq, bias, c["O"]), fpath
        return engine.quantlinear_float_input(xt, wp, _t(c["wd"]), sw, zw, bias), fpath
    xp, sx, zx = _t(c["xp"]), _t(c["sx"]), _t(c["zx"])
    xq, wq = capi.qparam(xp, c["ab"], c["asgn"], sx, zx), capi.qparam(wp, c["wb"], c["wsgn"], sw, zw)
    path = capi.linear_path(xq, wq, c["B"], c["K"], c["O"])
    if via_capi:
        return capi.quantlinear(xq, wq, bias, c["B"], c["K"], c["O"]), path
    return engine.quantlinear(xp, _t(c["xd"]), sx, zx, wp, _t(c["wd"]), sw, zw, bias), path


SWEEP = [(1, 16, 1), (5, 48, 300), (130, 64, 257), (257, 80, 33), (64, 768, 130), (129, 3072, 96), (40, 100, 70),
         (3, 7, 5), (200, 1024, 512)]


@pytest.mark.parametrize("via_capi", [False, True])
def test_random_sweep_vs_oracle(engine, via_capi):
    rng = np.random.RandomState(5)
    quant = [(8, 1, 8, 1), (8, 0, 8, 0), (8, 1, 8, 0), (4, 1, 4, 1), (3, 1, 5, 0), (8, 1, 0, 0), (4, 0, 0, 0), (8, 0, 0, 0)]
    seen_f32_mfma = []
    k = 0
    for shp in SWEEP:
        for (wb, wsgn, ab, asgn) in quant:
            k += 1
            c = _random_case(rng, *shp, wb, wsgn, ab, asgn, w_pc=k % 2 == 0, a_pr=k % 3 == 0, zeros=k % 4 != 0, bias=k % 5 != 0)
            y, path = _run(engine, c, via_capi)
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            assert got.shape == c["o32"].shape
            _close(got, c["o64"], c["o32"], "shape %s quant %s" % (shp, (wb, wsgn, ab, asgn)), c["fma"])
            if path == 0:
                assert np.array_equal(got, c["fma"]), "fp32 kernel not bit-exact: %s %s" % (shp, (wb, wsgn, ab, asgn))
            elif path == 2:
                assert wb == 8 and ab == 0 and shp[1] % 32 == 0
                seen_f32_mfma.append(shp)
            else:
                assert wb == 8 and ab == 8 and shp[1] % 16 == 0
    assert len(seen_f32_mfma) >= 6      # quantlinear_float_input ran on the matrix cores for the 8-bit-weight, K % 32 == 0 shapes


def test_float_input_mfma_vit_shapes(engine, monkeypatch):
    """quantlinear_float_input on the bf16 x 3 MFMA kernel at ViT-B/16 shapes (the 'float route' of a packed Linear): absolute
    1e-5 against the float64-exact value at the headline scales, ragged rows / columns, asymmetric weights; and the same
    problem on the fp32 chain kernel (QE_LIN_F32_MFMA=0) stays bit-identical to the oracle's fused chain."""
    rng = np.random.RandomState(23)
    for (B, K, O, zeros) in [(197, 768, 768, False), (130, 768, 3072, True), (70, 3072, 768, False), (256, 768, 1000, True), (1, 32, 1, True)]:
        c = _random_case(rng, B, K, O, 8, 1, 0, 0, w_pc=True, a_pr=False, zeros=zeros, bias=True)
        y, path = _run(engine, c, True)
        assert path == 2
        err = np.abs(y.cpu().numpy().astype(np.float64) - c["o64"]).max()
        assert err <= max(1e-5, np.abs(c["o32"].astype(np.float64) - c["o64"]).max()), (B, K, O, err)   # the conv rule: no headroom factor
    monkeypatch.setenv("QE_LIN_F32_MFMA", "0")
    capi.reload_env()
    y0, path0 = _run(engine, c, True)
    assert path0 == 0 and np.array_equal(y0.cpu().numpy(), c["fma"])
    monkeypatch.delenv("QE_LIN_F32_MFMA")
    capi.reload_env()


def test_vit_shapes_and_row_independence(engine):
    """ViT-B/16 layer shapes (K, O) at a reduced token count vs the oracle, then the full 50,432-row problem
    (256 images x 197 tokens) through a size-independent property: every row block equals the same rows computed
    on their own."""
    rng = np.random.RandomState(9)
    for (K, O) in [(768, 768), (768, 3072), (3072, 768), (768, 1000)]:
        c = _random_case(rng, 197, K, O, 8, 1, 8, 1, w_pc=True, a_pr=False, zeros=False, bias=True)
        y, path = _run(engine, c, True)
        assert path == 1
        # north_star's bar is ABSOLUTE 1e-5 at the headline operand scales (s_x = 2e-3, s_w ~ 5e-4: |out| is O(1))
        assert np.abs(y.cpu().numpy().astype(np.float64) - c["o64"]).max() <= 1e-5, (K, O)
    B, K, O = 256 * 197, 768, 768
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    qx = torch.randint(-128, 128, (B, K), generator=g, device=DEV, dtype=torch.int16)
    qw = torch.randint(-128, 128, (O, K), generator=g, device=DEV, dtype=torch.int16)
    xp, xd = engine.tpack(qx, 8, True)
    wp, wd = engine.tpack(qw, 8, True)
    sx = torch.rand(B, generator=g, device=DEV) * 2e-3 + 1e-3
    zx = torch.rand(B, generator=g, device=DEV) * 4 - 2
    sw = torch.rand(O, generator=g, device=DEV) * 5e-4 + 2.5e-4
    zw = torch.zeros(O, device=DEV)
    bias = torch.randn(O, generator=g, device=DEV) * 0.1
    y = engine.quantlinear(xp, xd, sx, zx, wp, wd, sw, zw, bias)
    assert tuple(y.shape) == (B, O)
    for r0 in (0, 12345, B - 130):
        rows = slice(r0, r0 + 130)
        xq, xdq = engine.tpack(qx[rows].contiguous(), 8, True)
        ys = engine.quantlinear(xq, xdq, sx[rows].contiguous(), zx[rows].contiguous(), wp, wd, sw, zw, bias)
        assert torch.equal(ys, y[rows])
    # one row block against the oracle
    r = slice(777, 777 + 40)
    o64 = oracle.quantlinear(*[t.cpu().numpy() for t in engine.tpack(qx[r].contiguous(), 8, True)],
                             sx[r].cpu().numpy(), zx[r].cpu().numpy(), wp.cpu().numpy(), wd.cpu().numpy(),
                             sw.cpu().numpy(), zw.cpu().numpy(), bias.cpu().numpy(), mode="f64", return_f64=True)[1]
    # north_star's bar is ABSOLUTE: 1e-5 at these operand scales (|out| stays below 1.5 here)
    assert np.abs(y[r].cpu().numpy().astype(np.float64) - o64).max() <= 1e-5


def test_error_messages(engine):
    one = torch.ones(1, device=DEV)
    xp, xd = engine.tpack(torch.zeros(2, 16, device=DEV), 8, True)
    wp, wd = engine.tpack(torch.zeros(3, 32, device=DEV), 8, True)
    with pytest.raises(RuntimeError, match="Input and weight do not match"):
        engine.quantlinear(xp, xd, one, one, wp, wd, one, one, None)
    wp, wd = engine.tpack(torch.zeros(3, 16, device=DEV), 8, True)
    with pytest.raises(RuntimeError, match="Weight and bias do not match"):
        engine.quantlinear(xp, xd, one, one, wp, wd, one, one, torch.zeros(4, device=DEV))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        engine.quantlinear(xp.cpu(), xd, one, one, wp, wd, one, one, None)
    with pytest.raises(RuntimeError, match="input must be a float tensor"):
        engine.quantlinear_float_input(torch.zeros(2, 16, device=DEV, dtype=torch.float64), wp, wd, one, one, None)
    with pytest.raises(RuntimeError, match="batch_size elements"):
        engine.quantlinear(xp, xd, torch.ones(5, device=DEV), torch.ones(5, device=DEV), wp, wd, one, one, None)
    # 0-dim scales are expanded by the reference (quantlinear.cu:276-290)
    s0 = torch.tensor(0.5, device=DEV)
    y = engine.quantlinear(xp, xd, s0, 0 * s0, wp, wd, s0, 0 * s0, torch.ones(3, device=DEV))
    assert tuple(y.shape) == (2, 3) and torch.equal(y, torch.ones(2, 3, device=DEV))


def test_deep_reductions_leave_the_int32_kernel():
    """K >= 2^17 products of up to 2^14 each could overflow the MFMA kernel's int32 accumulators: those problems keep the
    fp32 kernel (the reference accumulates in fp32, quantlinear.cu:96-127)."""
    dev = torch.device("cuda")
    one = torch.ones(1, device=dev)
    buf = torch.zeros(4 * (1 << 17), dtype=torch.uint8, device=dev)
    xq, wq = capi.qparam(buf, 8, True, one, one), capi.qparam(buf, 8, True, one, one)
    assert capi.linear_path(xq, wq, 4, (1 << 17) - 64, 4) == 1
    assert capi.linear_path(xq, wq, 4, 1 << 17, 4) == 0


def test_big_tile_kernel_forced(engine, monkeypatch):
    """linear_mfma8_kernel (320 x 256 tiles, 8 waves) forced onto the sweep shapes it is eligible for (ragged row and column
    counts, per-row activation scales, asymmetric operands, 1 to 48 stages): against the oracle and, bit for bit, against
    the 128 x 256 kernel (same integer sums, same epilogue operation order)."""
    from quantize_amd import capi as _capi
    rng = np.random.RandomState(11)
    k = 0
    for shp in [(330, 768, 256), (129, 3072, 256), (200, 1024, 512), (700, 128, 512), (321, 256, 256), (640, 3072, 256), (64, 768, 130), (1, 128, 4)]:
        for (wsgn, asgn) in [(1, 1), (0, 0), (1, 0)]:
            k += 1
            c = _random_case(rng, *shp, 8, wsgn, 8, asgn, w_pc=k % 2 == 0, a_pr=k % 3 != 0, zeros=k % 4 != 0, bias=k % 5 != 0)
            monkeypatch.setenv("QE_LIN8", "0")
            _capi.reload_env()
            y4, _ = _run(engine, c, True)
            torch.cuda.synchronize()
            for form in ("1", "2"):                  # 8 waves / 320-row tiles; 4 waves / 160-row tiles, one stage buffer
                monkeypatch.setenv("QE_LIN8", form)
                _capi.reload_env()
                y8, path = _run(engine, c, True)
                torch.cuda.synchronize()
                assert path == 1
                _close(y8.cpu().numpy(), c["o64"], c["o32"], "big tile %s form %s" % (shp, form), c["fma"])
                assert torch.equal(y8, y4), (shp, form)
    monkeypatch.delenv("QE_LIN8")
    _capi.reload_env()
