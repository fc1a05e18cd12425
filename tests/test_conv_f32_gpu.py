"""GPU: quantconv2d_float_input on the bf16 MFMA kernel (qe_conv_f32.hip: exact 3-way bf16 split of the fp32 activations
x integer weight codes) vs the oracle.  Bar: |out - exact64| <= max(1e-5, |reference fp32 chain - exact64|) (the
reference's own sequential chain, as written and contracted), and plain 1e-5 on the ResNet-50 layer shapes at the
headline weight scales.  The order-preserving VALU kernel stays bit-identical to the contracted chain."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import conv_tolerance
from quantize_amd import capi, resnet50
from test_conv_gpu import _random_case, _run_case, _assert_conv_close, _t, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, IC, H, W, OC, K, stride, pad
    (2, 64, 56, 56, 64, 3, 1, 1),
    (2, 64, 28, 28, 130, 3, 1, 1),      # ragged output-channel tile
    (3, 128, 14, 14, 256, 3, 1, 1),
    (5, 160, 7, 7, 96, 3, 1, 1),        # several images per tile, last group partial, padded last channel group
    (2, 96, 30, 30, 72, 3, 2, 1),       # stride 2
    (2, 64, 56, 56, 256, 1, 1, 0),
    (2, 256, 56, 56, 64, 1, 1, 0),
    (3, 72, 28, 28, 40, 1, 2, 0),       # strided 1x1 (sampled rows / columns only), IC % 16 != 0
    (2, 40, 17, 19, 24, 5, 1, 2),       # 5x5, odd plane, quads crossing the row end
    (1, 16, 9, 9, 8, 3, 1, 0),          # no padding
    (9, 512, 7, 7, 128, 1, 1, 0),
    (2, 8, 12, 12, 16, 3, 1, 1),        # the smallest channel depth of the 16-channel k-step kernel
    # IC <= 4: the stem kernel (K = kh x [kw][ic])
    (2, 3, 64, 64, 64, 7, 2, 3),        # ResNet stem at 64x64: 64-channel workgroups, 14 column slots
    (2, 3, 33, 37, 24, 7, 2, 3),        # odd plane, partial oc tile, quads crossing the row end
    (2, 4, 20, 24, 48, 5, 1, 2),        # 4 input channels, 5x5, stride 1 (odd fragment columns)
    (2, 3, 32, 32, 130, 3, 1, 1),       # CIFAR stem: 3x3 / 1, 128-channel workgroups with a ragged second tile
    (3, 1, 28, 28, 16, 5, 1, 2),        # one input channel
    (1, 3, 224, 224, 64, 7, 2, 3),      # the ResNet-50 stem at its real size
]


def test_float_input_mfma_vs_oracle(engine):
    rng = np.random.RandomState(321)
    n_mfma = 0
    for shp in SHAPES:
        for (wb, wsgn, zeros) in [(8, 1, False), (8, 0, True), (4, 1, True), (3, 0, False)]:
            for via_capi in (False, True):
                case = _random_case(rng, *shp, wb, wsgn, 0, 0, w_pc=True, a_pc=False, zeros=zeros, bias=True)
                y, o32, o64 = _run_case(engine, case, via_capi=via_capi)
                n_mfma += int(case["path"] == 2)
                _assert_conv_close(y, o64, o32, "f32 %s w%d sgn%d zeros=%s capi=%s" % (shp, wb, wsgn, zeros, via_capi), case["fma"])
    assert n_mfma == len(SHAPES) * 8


def test_resnet50_shapes_1e5():
    """All 23 ResNet-50 conv shapes (the stem on its own K = kh x [kw][ic] kernel), N = 1, headline weight scales, fp32
    activations ~ N(0, 0.25): plain 1e-5 absolute against the float64-exact value."""
    rng = np.random.RandomState(8)
    seen = set()
    for layer in resnet50.conv_layers():
        sig = tuple(layer[1:])
        if sig in seen:
            continue
        seen.add(sig)
        qw = rng.randint(-128, 128, size=(layer.OC, layer.IC, layer.K, layer.K))
        sw = rng.uniform(2.5e-4, 7.5e-4, size=layer.OC).astype(np.float32)
        zw = np.zeros(layer.OC, np.float32)
        bias = rng.normal(0, 0.1, size=layer.OC).astype(np.float32)
        xf = rng.normal(0, 0.5, size=(1, layer.IC, layer.H, layer.H)).astype(np.float32)
        wp, wd = oracle.tpack(qw, 8, True)
        sh = capi.conv_shape(1, layer.IC, layer.H, layer.H, layer.OC, layer.K, layer.K, layer.stride, layer.pad)
        wq = capi.qparam(_t(wp), 8, 1, _t(sw), _t(zw))
        assert capi.float_input_path(sh, wq) == 1, layer.name
        y = capi.quantconv2d_float_input(_t(xf), wq, _t(bias), sh).cpu().numpy()
        _, o64 = oracle.quantconv2d_float_input(xf, wp, wd, sw, zw, bias, layer.stride, layer.pad, mode="f64", return_f64=True)
        assert np.abs(y.astype(np.float64) - o64).max() <= 1e-5, layer.name
    assert len(seen) == 23


def test_prepared_and_valu_forms():
    rng = np.random.RandomState(12)
    for shp in [(2, 64, 28, 28, 96, 3, 1, 1), (3, 128, 14, 14, 64, 1, 1, 0)]:
        case = _random_case(rng, *shp, 8, 1, 0, 0, w_pc=True, a_pc=False, zeros=True, bias=True)
        wp, wd, sw, zw = case["w"]
        xf = _t(case["xf"])
        N, IC, H, W = case["xf"].shape
        sh = capi.conv_shape(N, IC, H, W, int(wd[2]), int(wd[4]), int(wd[5]), case["stride"], case["pad"])
        wq = capi.qparam(_t(wp), int(wd[0]), int(wd[1]), _t(sw), _t(zw))
        bias = _t(case["bias"])
        y0 = capi.quantconv2d_float_input(xf, wq, bias, sh)                      # prepare + run
        prepared = capi.conv_f32_prepare(wq, bias, sh)
        assert prepared.numel() > 0
        y1 = capi.quantconv2d_float_input_prepared(xf, wq, bias, sh, prepared)  # run on kept tables
        assert torch.equal(y0, y1)
        # the plain entry point keeps the order-preserving kernel: bit-identical to the reference's contracted chain
        yv = capi.quantconv2d_float_input(xf, wq, bias, sh, mfma=False).cpu().numpy()
        fma = oracle.quantconv2d_float_input(case["xf"], wp, wd, sw, zw, case["bias"], case["stride"], case["pad"], mode="fp32_fma")
        assert np.array_equal(yv, fma)


def test_env_switch_keeps_valu_kernel(engine):
    rng = np.random.RandomState(13)
    old = os.environ.get("QE_F32_MFMA")
    os.environ["QE_F32_MFMA"] = "0"
    try:
        case = _random_case(rng, 2, 64, 14, 14, 48, 3, 1, 1, 8, 1, 0, 0, w_pc=True, a_pc=False, zeros=True, bias=True)
        y, o32, o64 = _run_case(engine, case, via_capi=False)
        assert case["path"] == 0 and np.array_equal(y, case["fma"])
    finally:
        if old is None:
            os.environ.pop("QE_F32_MFMA", None)
        else:
            os.environ["QE_F32_MFMA"] = old


def test_non_finite_activations_declared_behaviour(engine):
    """Declared in include/quant_engine.h: an output that depends on a non-finite activation (+-inf, NaN) is non-finite -- the
    bf16 x 3 split turns +-inf into NaN in the remainder terms where the reference's fmaf chain may keep +-inf -- and every
    other output is exactly what it is without that activation.  Checked on the MFMA kernels of quantconv2d_float_input
    (3x3 with padding, 1x1) and of quantlinear_float_input."""
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    for (N, IC, H, OC, K, s, p) in [(2, 64, 14, 96, 3, 1, 1), (2, 128, 14, 64, 1, 1, 0)]:
        x = torch.randn(N, IC, H, H, generator=g, device="cuda")
        qw = torch.randint(-128, 128, (OC, IC, K, K), generator=g, device="cuda", dtype=torch.int16)
        wp, wd = engine.tpack(qw, 8, True)
        sw = (torch.rand(OC, generator=g, device="cuda") * 5e-4 + 2.5e-4).reshape(OC, 1, 1, 1)
        zw = torch.zeros(OC, 1, 1, 1, device="cuda")
        bad = [(0, 3, 5, 6, float("inf")), (1, 10, 0, 13, float("-inf")), (1, 20, 9, 2, float("nan"))]
        xb, xc = x.clone(), x.clone()
        for (n, c, h, w, v) in bad:
            xb[n, c, h, w] = v
            xc[n, c, h, w] = 0.0
        yb = engine.quantconv2d_float_input(xb, wp, wd, sw, zw, None, s, p)
        yc = engine.quantconv2d_float_input(xc, wp, wd, sw, zw, None, s, p)
        OH = yb.shape[2]
        hit = torch.zeros(N, OH, OH, dtype=torch.bool, device="cuda")
        for (n, c, h, w, v) in bad:
            for oh in range(OH):
                for ow in range(OH):
                    if 0 <= h - (oh * s - p) < K and 0 <= w - (ow * s - p) < K:
                        hit[n, oh, ow] = True
        hit = hit[:, None].expand_as(yb)
        assert hit.any() and not torch.isfinite(yb[hit]).any(), (IC, K)
        assert torch.isfinite(yb[~hit]).all() and torch.equal(yb[~hit], yc[~hit]), (IC, K)
    # linear: a non-finite activation poisons its whole output row, nothing else
    B, Kd, O = 130, 768, 256
    x = torch.randn(B, Kd, generator=g, device="cuda")
    qw = torch.randint(-128, 128, (O, Kd), generator=g, device="cuda", dtype=torch.int16)
    wp, wd = engine.tpack(qw, 8, True)
    sw = torch.rand(O, generator=g, device="cuda") * 5e-4 + 2.5e-4
    zw = torch.zeros(O, device="cuda")
    assert capi.linear_float_input_path(x, capi.qparam(wp, 8, 1, sw, zw), B, Kd, O) == 1
    xb, xc = x.clone(), x.clone()
    for (r, k, v) in [(3, 5, float("inf")), (77, 700, float("-inf")), (129, 31, float("nan"))]:
        xb[r, k] = v
        xc[r, k] = 0.0
    yb = engine.quantlinear_float_input(xb, wp, wd, sw, zw, None)
    yc = engine.quantlinear_float_input(xc, wp, wd, sw, zw, None)
    rows = torch.zeros(B, dtype=torch.bool, device="cuda")
    rows[[3, 77, 129]] = True
    assert not torch.isfinite(yb[rows]).any()
    assert torch.isfinite(yb[~rows]).all() and torch.equal(yb[~rows], yc[~rows])
