"""GPU: weights kept prepared across calls (reference quantconv2d.py:187-192, :230-233).
qe_conv_prepare + qe_quantconv2d_prepared must be BIT-identical to qe_quantconv2d on every kernel family, and the torch
module's host-side caches (parsed des, prepared tables) must hit on repeated calls and miss after the weights change."""
import numpy as np
import pytest
import torch

import oracle
from quantize_amd import capi
from test_conv_gpu import _random_case, _t, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    (2, 64, 56, 56, 64, 3, 1, 1),      # 3x3 two-strip kernel
    (3, 128, 28, 28, 128, 3, 1, 1),    # 3x3 halo kernel
    (2, 128, 56, 56, 130, 3, 2, 1),    # stride-2 3x3, ragged output-channel tile
    (5, 512, 7, 7, 512, 3, 1, 1),      # warp-specialised 3x3 on 7x7 maps, partial image group
    (2, 3, 64, 64, 64, 7, 2, 3),       # stem
    (2, 256, 14, 14, 256, 3, 1, 1),    # 3x3 on 14x14
    (2, 96, 28, 28, 72, 1, 1, 0),      # 1x1 with IC % 64 != 0: register-staged flat kernel, nothing to prepare for 8-bit weights
    (2, 256, 28, 28, 140, 1, 2, 0),    # strided 1x1: gather + flat
    (4, 512, 7, 7, 256, 1, 1, 0),      # 7x7 1x1 (DMA ring kernel)
]


@pytest.mark.parametrize("quant", [(8, 1, 8, 1, False), (8, 1, 8, 0, True), (4, 1, 8, 1, True), (4, 1, 4, 0, True)])
def test_prepared_equals_per_call(quant):
    wb, wsgn, ab, asgn, zeros = quant
    rng = np.random.RandomState(77 + wb + ab)
    n_prepared = 0
    for shp in SHAPES:
        case = _random_case(rng, *shp, wb, wsgn, ab, asgn, w_pc=True, a_pc=False, zeros=zeros, bias=True)
        wp, wd, sw, zw = case["w"]
        xp, xd, sx, zx = case["x"]
        N, IC, H, W = [int(v) for v in xd[2:6]]
        sh = capi.conv_shape(N, IC, H, W, int(wd[2]), int(wd[4]), int(wd[5]), case["stride"], case["pad"])
        xq = capi.qparam(_t(xp), int(xd[0]), int(xd[1]), _t(sx), _t(zx))
        wq = capi.qparam(_t(wp), int(wd[0]), int(wd[1]), _t(sw), _t(zw))
        bias = _t(case["bias"])
        y0 = capi.quantconv2d(xq, wq, bias, sh)
        prepared = capi.conv_prepare(wq, bias, sh, ab)
        n_prepared += int(prepared.numel() > 0)
        y1 = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)
        y2 = capi.quantconv2d_prepared(xq, wq, bias, sh, prepared)      # the tables are not consumed
        assert torch.equal(y0, y1) and torch.equal(y0, y2), (shp, quant)
        # a different activation scale on the same prepared tables (they are x-independent)
        sx2 = _t(np.array([3.1e-3], np.float32))
        xq2 = capi.qparam(_t(xp), int(xd[0]), int(xd[1]), sx2, _t(zx))
        assert torch.equal(capi.quantconv2d(xq2, wq, bias, sh), capi.quantconv2d_prepared(xq2, wq, bias, sh, prepared)), shp
    assert n_prepared >= 5


def test_prepare_argument_checks():
    L = capi.lib()
    import ctypes
    sh = capi.conv_shape(1, 64, 8, 8, 64, 3, 3, 1, 1)
    w = torch.zeros(64 * 64 * 9, dtype=torch.uint8, device="cuda:0")
    one = torch.ones(1, device="cuda:0")
    wq = capi.qparam(w, 8, 1, one, one)
    need = int(L.qe_conv_prepared_bytes(ctypes.byref(sh), 8, 8))
    assert need > 0
    small = torch.empty(need - 16, dtype=torch.uint8, device="cuda:0")
    assert L.qe_conv_prepare(ctypes.byref(wq), None, ctypes.byref(sh), 8, small.data_ptr(), small.numel(), None) == 6   # workspace too small
    assert L.qe_conv_prepare(ctypes.byref(wq), None, ctypes.byref(sh), 9, small.data_ptr(), need, None) == 1           # n_bits
    sh1 = capi.conv_shape(1, 64, 8, 8, 64, 1, 1, 1, 0)
    assert int(L.qe_conv_prepared_bytes(ctypes.byref(sh1), 8, 8)) == 0      # 8-bit 1x1: the kernel reads the packed weights itself
    assert L.qe_conv_prepare(ctypes.byref(wq), None, ctypes.byref(sh1), 8, None, 0, None) == 0


def test_module_caches_hit_and_invalidate(engine):
    import quant_engine
    quant_engine.clear_cache()
    rng = np.random.RandomState(5)
    case = _random_case(rng, 2, 64, 14, 14, 96, 3, 1, 1, 8, 1, 8, 1, w_pc=True, a_pc=False, zeros=False, bias=True)
    wp, wd, sw, zw = case["w"]
    xp, xd, sx, zx = case["x"]
    t = dict(xp=_t(xp), xd=_t(xd), sx=_t(sx), zx=_t(zx), wp=_t(wp), wd=_t(wd), sw=_t(sw).reshape(-1, 1, 1, 1),
             zw=_t(zw).reshape(-1, 1, 1, 1), b=_t(case["bias"]))
    call = lambda: engine.quantconv2d(t["xp"], t["xd"], t["sx"], t["zx"], t["wp"], t["wd"], t["sw"], t["zw"], t["b"], 1, 1)
    sb = quant_engine.cache_stats()             # the counters are cumulative over the process: compare differences
    y0 = call()
    s0 = quant_engine.cache_stats()
    assert s0[1] - sb[1] == 2 and s0[3] - sb[3] == 1 and s0[4:] == [2, 1]   # two descriptions parsed, one weight set prepared
    y1 = call()
    s1 = quant_engine.cache_stats()
    assert s1[0] == s0[0] + 2 and s1[2] == s0[2] + 1 and s1[1] == s0[1] and s1[3] == s0[3]   # all hits, no new misses
    assert torch.equal(y0, y1)
    _, ref = oracle.quantconv2d(xp, xd, sx, zx, wp, wd, sw, zw, case["bias"], 1, 1, mode="f64", return_f64=True)
    assert np.abs(y1.cpu().numpy().astype(np.float64) - ref).max() <= 1e-5
    # in-place change of the packed weights: version bump -> re-prepared, result follows the new weights
    t["wp"].copy_(torch.flip(t["wp"], dims=[0]))
    y2 = call()
    s2 = quant_engine.cache_stats()
    assert s2[3] == s1[3] + 1
    wp2 = np.ascontiguousarray(wp[::-1])
    _, ref2 = oracle.quantconv2d(xp, xd, sx, zx, wp2, wd, sw, zw, case["bias"], 1, 1, mode="f64", return_f64=True)
    assert np.abs(y2.cpu().numpy().astype(np.float64) - ref2).max() <= 1e-5
    # in-place change of a scale as well
    t["sw"].mul_(2.0)
    y3 = call()
    assert quant_engine.cache_stats()[3] == s2[3] + 1
    assert np.abs(y3.cpu().numpy().astype(np.float64) - (2.0 * (ref2 - case["bias"].reshape(1, -1, 1, 1)) + case["bias"].reshape(1, -1, 1, 1))).max() <= 2e-5
    quant_engine.clear_cache()
    assert quant_engine.cache_stats()[4:] == [0, 0]


def test_prepared_entry_is_shared_across_batch_sizes(engine):
    """The prepared tables do not depend on the batch size: a layer called with alternating batch sizes (7, 3, 7, ...)
    re-uses ONE cache entry (keyed on qe_conv_prepared_layout, not on the whole problem shape) and every result matches
    the oracle."""
    import quant_engine
    quant_engine.clear_cache()
    rng = np.random.RandomState(91)
    big = _random_case(rng, 7, 64, 14, 14, 96, 3, 1, 1, 8, 1, 8, 1, w_pc=True, a_pc=False, zeros=False, bias=True)
    wp, wd, sw, zw = big["w"]
    xp, xd, sx, zx = big["x"]
    qx = oracle.tunpack(xp, xd)
    w = dict(wp=_t(wp), wd=_t(wd), sw=_t(sw).reshape(-1, 1, 1, 1), zw=_t(zw).reshape(-1, 1, 1, 1), b=_t(big["bias"]))
    sh7 = capi.conv_shape(7, 64, 14, 14, 96, 3, 3, 1, 1)
    sh3 = capi.conv_shape(3, 64, 14, 14, 96, 3, 3, 1, 1)
    assert capi.conv_prepared_layout(sh7, 8, 8) == capi.conv_prepared_layout(sh3, 8, 8) != 0
    misses0 = quant_engine.cache_stats()[3]
    for n in (7, 3, 7, 3, 1):
        xpn, xdn = oracle.tpack(qx[:n], 8, 1)
        y = engine.quantconv2d(_t(xpn), _t(xdn), _t(sx), _t(zx), w["wp"], w["wd"], w["sw"], w["zw"], w["b"], 1, 1)
        _, ref = oracle.quantconv2d(xpn, xdn, sx, zx, wp, wd, sw, zw, big["bias"], 1, 1, mode="f64", return_f64=True)
        assert np.abs(y.cpu().numpy().astype(np.float64) - ref).max() <= 1e-5, n
    assert quant_engine.cache_stats()[3] == misses0 + 1      # prepared once for all five calls
