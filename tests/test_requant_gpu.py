"""GPU: quantconv2d with the consumer's activation quantiser fused into the epilogue (qe_quantconv2d_requant_prepared,
SURVEY.md section 8 row f-2) must equal qe_quantize_pack(qe_quantconv2d_prepared(...)) BIT FOR BIT, and that must equal
the oracle's tpack of round(y / s - z).clamp(qmin, qmax) computed on the host from the engine's own fp32 y -- for every
kernel family (flat, stride-2 flat, 7x7 small-plane, halo, two-strip, warp-specialised, stem), ragged tiles, symmetric
and asymmetric operands, signed and unsigned codes, plus the two-pass route (sub-8-bit codes, per-channel scale)."""
import numpy as np
import pytest
import torch

import oracle
from quantize_amd import capi
from test_conv_gpu import _random_case, _t, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, IC, H, W, OC, K, stride, pad
    (2, 64, 56, 56, 256, 1, 1, 0),      # flat, 224-pixel tiles
    (2, 256, 56, 56, 64, 1, 1, 0),      # flat, 2x2 wave layout (64 output channels)
    (3, 128, 28, 28, 130, 1, 1, 0),     # flat, 160-pixel tiles with a partial last tile, ragged output-channel tile
    (2, 256, 14, 14, 1024, 1, 1, 0),    # 196-pixel planes (rows only 4-byte aligned)
    (2, 128, 56, 56, 160, 1, 2, 0),     # stride-2 flat
    (3, 256, 28, 28, 140, 1, 2, 0),     # stride 2: gathered, then flat
    (5, 512, 7, 7, 200, 1, 1, 0),       # 7x7 small-plane kernel (49-byte planes), partial image group, ragged channels
    (4, 2048, 7, 7, 512, 1, 1, 0),
    (2, 64, 56, 56, 64, 3, 1, 1),       # two-strip 3x3
    (2, 256, 14, 14, 256, 3, 1, 1),
    (2, 128, 56, 56, 128, 3, 2, 1),     # halo kernel, stride 2
    (5, 512, 7, 7, 512, 3, 1, 1),       # warp-specialised
    (2, 3, 64, 64, 64, 7, 2, 3),        # stem
    (2, 33, 9, 9, 31, 5, 1, 2),         # generic taps, tiny
    # the resident-tile 1x1 kernel's own re-quantising epilogue (qe_conv_pwr.hip, RQ instances)
    (2, 128, 28, 28, 512, 1, 1, 0),     # 196-byte row pieces of 784-byte planes (last 16-byte piece of a row shifted back)
    (1, 256, 28, 28, 512, 1, 1, 0),     # 8 waves
    (3, 256, 56, 56, 512, 1, 2, 0),     # its stride-2 form
    (2, 64, 28, 64, 256, 1, 2, 0),      # stride 2, 224-byte row pieces
    (40, 64, 14, 14, 192, 1, 1, 0),     # whole-plane tiles: a strip's 32 planes are one 6272-byte run; more tiles than XCDs
    (3, 128, 14, 14, 160, 1, 1, 0),     # 5 strips on 4 waves
    (4, 512, 7, 7, 2048, 1, 1, 0),      # the 7x7-plane form: a strip's 32 planes x 49 codes of an image are one 1568-byte run
    (8, 256, 7, 7, 768, 1, 1, 0),       # 4 images per tile, 3 strips per wave
    (8, 128, 7, 7, 1024, 1, 1, 0),
]


def _case_tensors(case):
    wp, wd, sw, zw = case["w"]
    xp, xd, sx, zx = case["x"]
    N, IC, H, W = [int(v) for v in xd[2:6]]
    sh = capi.conv_shape(N, IC, H, W, int(wd[2]), int(wd[4]), int(wd[5]), case["stride"], case["pad"])
    xq = capi.qparam(_t(xp), int(xd[0]), int(xd[1]), _t(sx), _t(zx))
    wq = capi.qparam(_t(wp), int(wd[0]), int(wd[1]), _t(sw), _t(zw))
    bias = None if case["bias"] is None else _t(case["bias"])
    return sh, xq, wq, bias


@pytest.fixture
def fresh_env():
    """The library reads QE_* once per process: re-read after the test's monkeypatch is undone."""
    yield
    import os
    os.environ.pop("QE_PWR_RQ", None)
    capi.reload_env()


@pytest.mark.parametrize("sign,pwr_rq", [(True, "1"), (False, "1"), (True, "0")])
def test_requant_equals_conv_then_quantize_pack(engine, sign, pwr_rq, monkeypatch, fresh_env):
    """pwr_rq = 0: the resident-tile kernel's re-quantising instances off -- the flat kernels' fused epilogue against the
    fp32 output of the resident-tile kernel (two kernels, one arithmetic)."""
    monkeypatch.setenv("QE_PWR_RQ", pwr_rq)
    capi.reload_env()
    rng = np.random.RandomState(77 + sign)
    for shp in SHAPES:
        for zeros in (False, True):
            case = _random_case(rng, *shp, 8, 1, 8, 0 if zeros else 1, w_pc=True, a_pc=False, zeros=zeros, bias=True)
            sh, xq, wq, bias = _case_tensors(case)
            prepared = capi.conv_prepare(wq, bias, sh, 8)
            y = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)
            torch.cuda.synchronize()
            amax = float(y.abs().max())
            qmin, qmax = (-128.0, 127.0) if sign else (0.0, 255.0)
            s = torch.tensor([amax / 100.0], device="cuda")          # clips a few percent of the values
            z = torch.tensor([0.37 if sign else -117.25], device="cuda")
            rq = capi.requant(s, z, qmin, qmax, 8, sign)
            assert capi.requant_path(sh, xq, wq, rq) == 1, shp
            got, status = capi.quantconv2d_requant_prepared(xq, wq, bias, sh, prepared, rq)
            ref, st2 = capi.quantize_pack(y, s, z, qmin, qmax, 8, sign)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and int(st2.item()) == 0
            g, r = got.cpu().numpy(), ref.cpu().numpy()
            assert g.shape == r.shape
            bad = np.nonzero(g != r)[0]
            assert bad.size == 0, "%s zeros=%s: %d of %d codes differ, first at %d (%d vs %d)" % (
                shp, zeros, bad.size, g.size, bad[0], g[bad[0]], r[bad[0]])
            # the same codes from the host arithmetic of the oracle on the engine's fp32 output
            yq = np.clip(np.rint(y.cpu().numpy() / np.float32(s.item()) - np.float32(z.item())), qmin, qmax)
            op, _ = oracle.tpack(yq.astype(np.int64), 8, sign)
            assert np.array_equal(g, op), shp


def test_requant_two_pass_routes(engine):
    """4-bit codes and per-channel output scales take conv + quantise+pack inside the call: same bits, path 0."""
    rng = np.random.RandomState(5)
    for shp, bits, per_ch in [((2, 64, 14, 14, 96, 1, 1, 0), 4, False), ((2, 64, 14, 14, 96, 3, 1, 1), 8, True),
                              ((3, 128, 7, 7, 130, 1, 1, 0), 5, True)]:
        case = _random_case(rng, *shp, 8, 1, 8, 1, w_pc=True, a_pc=False, zeros=False, bias=True)
        sh, xq, wq, bias = _case_tensors(case)
        prepared = capi.conv_prepare(wq, bias, sh, 8)
        y = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)
        OC = shp[4]
        n = OC if per_ch else 1
        s = (torch.rand(n, device="cuda") + 0.5) * float(y.abs().max()) / (1 << (bits - 1))
        z = torch.rand(n, device="cuda") - 0.5
        qmin, qmax = -float(1 << (bits - 1)), float((1 << (bits - 1)) - 1)
        rq = capi.requant(s, z, qmin, qmax, bits, True)
        assert capi.requant_path(sh, xq, wq, rq) == 0
        got, status = capi.quantconv2d_requant_prepared(xq, wq, bias, sh, prepared, rq)
        ref, _ = capi.quantize_pack(y, s, z, qmin, qmax, bits, True, inner=y.shape[2] * y.shape[3])
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and torch.equal(got, ref)


def test_requant_range_flag(engine):
    """qmax beyond the 8-bit code range trips the status flag exactly as qe_quantize_pack does (the codes are unspecified
    then: the reference raises "The input tensor is out of range.")."""
    rng = np.random.RandomState(6)
    case = _random_case(rng, 2, 64, 14, 14, 64, 1, 1, 0, 8, 1, 8, 1, w_pc=True, a_pc=False, zeros=False, bias=True)
    sh, xq, wq, bias = _case_tensors(case)
    prepared = capi.conv_prepare(wq, bias, sh, 8)
    y = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)
    s = torch.tensor([float(y.abs().max()) / 300.0], device="cuda")
    z = torch.zeros(1, device="cuda")
    rq = capi.requant(s, z, -300.0, 300.0, 8, True)
    got, status = capi.quantconv2d_requant_prepared(xq, wq, bias, sh, prepared, rq)
    ref, st2 = capi.quantize_pack(y, s, z, -300.0, 300.0, 8, True)
    torch.cuda.synchronize()
    assert int(status.item()) == 1 and int(st2.item()) == 1


def test_requant_division_free_quotient_is_exact(engine):
    """The epilogue replaces y / scale by Markstein's two-fma sequence on a reciprocal; that must give the codes of the
    IEEE division of qe_quantize_pack for ANY scale: random ones, scales whose significand is all ones (the case the
    theorem excludes: the kernel divides there), very small and very large ones, zero points that put many values on
    .5 boundaries, both kernel orientations (flat: lane = channel, halo: lane = pixel)."""
    rng = np.random.RandomState(99)
    for shp in [(2, 64, 28, 28, 128, 1, 1, 0), (2, 64, 14, 14, 128, 3, 1, 1)]:
        case = _random_case(rng, *shp, 8, 1, 8, 1, w_pc=True, a_pc=False, zeros=False, bias=True)
        sh, xq, wq, bias = _case_tensors(case)
        prepared = capi.conv_prepare(wq, bias, sh, 8)
        y = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)
        amax = float(y.abs().max())
        scales = [amax / d for d in rng.uniform(20, 400, size=24)]
        scales += [float(np.float32(amax / 90).view(np.uint32) | np.uint32(0x7fffff)) and
                   float((np.float32(amax / 90).view(np.uint32) | np.uint32(0x7fffff)).view(np.float32))]   # all-ones significand
        scales += [amax * 1e-6, amax * 1e3, 1e-30, 1e30, float(np.float32(2.0) ** -70), float(np.float32(2.0) ** 70)]
        for k, sc in enumerate(scales):
            s = torch.tensor([sc], device="cuda", dtype=torch.float32)
            z = torch.tensor([[0.0, 0.5, -3.25, 17.125][k % 4]], device="cuda")
            rq = capi.requant(s, z, -128.0, 127.0, 8, True)
            got, status = capi.quantconv2d_requant_prepared(xq, wq, bias, sh, prepared, rq)
            ref, st2 = capi.quantize_pack(y, s, z, -128.0, 127.0, 8, True)
            torch.cuda.synchronize()
            assert int(status.item()) == int(st2.item()) == 0
            assert torch.equal(got, ref), "scale %r zero %r: %d codes differ" % (sc, float(z), int((got != ref).sum()))


def test_requant_on_the_sweep_shapes(engine):
    """Every odd shape of the conv sweep (tiny planes, 1x1 pixels, 5x5 / 7x7 taps, ragged everything), 8- and 4-bit
    activations: fused or two-pass, the codes equal quantize_pack of the fp32 result."""
    from test_conv_gpu import SWEEP_SHAPES
    rng = np.random.RandomState(314)
    n_fused = 0
    for shp in SWEEP_SHAPES:
        for ab in (8, 4):
            case = _random_case(rng, *shp, 8, 1, ab, 0, w_pc=True, a_pc=False, zeros=True, bias=True)
            sh, xq, wq, bias = _case_tensors(case)
            prepared = capi.conv_prepare(wq, bias, sh, ab)
            y = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)
            s = (y.abs().max() / 90.0).reshape(1).clamp(min=1e-6)
            z = torch.tensor([-3.5], device="cuda")
            rq = capi.requant(s, z, 0.0, 255.0, 8, False)
            n_fused += capi.requant_path(sh, xq, wq, rq)
            got, status = capi.quantconv2d_requant_prepared(xq, wq, bias, sh, prepared, rq)
            ref, _ = capi.quantize_pack(y, s, z, 0.0, 255.0, 8, False)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and torch.equal(got, ref), (shp, ab, int((got != ref).sum()))
    assert n_fused >= 30
