"""GPU: the resident-tile 1x1 kernel (qe_conv_pwr.hip) vs the oracle: every instance (IC 64 / 128 / 256 x tiles of 224
and 196 pixels), planes that are one tile (14x14: the 4-byte patched tail of the tensor), several tiles (28x28, 56x56),
one to sixteen strips per wave, channel groups (QE_PWR_GROUPS), more tiles than XCDs, symmetric and asymmetric operands
(S_x / S_w terms), per-tensor weight scales, no bias -- each also with the kernel disabled (QE_PWR=0), and a batch-256
launch checked by batch independence."""
import os

import numpy as np
import pytest

from quantize_amd import capi
from test_conv_gpu import _random_case, _run_case, _assert_conv_close, engine  # noqa: F401

pytestmark = pytest.mark.gpu

SHAPES = [
    # N, IC, H, W, OC, K, stride, pad
    (2, 256, 14, 14, 1024, 1, 1, 0),    # ResNet-50 layer3 expansion: whole-plane tiles, 4 strips per wave (8 waves)
    (3, 256, 14, 14, 128, 1, 1, 0),     # fewer strips than waves: idle waves only stage the tile
    (1, 128, 14, 14, 160, 1, 1, 0),     # 5 strips on 4 waves (uneven)
    (2, 64, 14, 14, 256, 1, 1, 0),
    (2, 128, 28, 28, 512, 1, 1, 0),     # ResNet-50 layer2 expansion: 4 tiles of 196 pixels per plane
    (1, 256, 28, 28, 512, 1, 1, 0),     # the dense form of layer2's downsample branch
    (3, 256, 56, 56, 512, 1, 2, 0),     # ResNet-50 layer2.0.downsample: the stride-2 form (even columns of even rows, fetched once per tile)
    (2, 64, 56, 56, 256, 1, 2, 0),      # stride 2, IC = 64
    (1, 128, 56, 56, 512, 1, 2, 0),     # stride 2, IC = 128
    (2, 64, 28, 64, 256, 1, 2, 0),      # stride 2, 64-byte input rows (8 pieces), 14 x 32 output planes = 2 tiles of 224
    (1, 64, 56, 56, 256, 1, 1, 0),      # ResNet-50 layer1 expansion: 14 tiles of 224 pixels per plane
    (2, 256, 56, 56, 128, 1, 1, 0),     # layer2.0.conv1
    (40, 64, 14, 14, 192, 1, 1, 0),     # more tiles than XCDs
    (1, 128, 28, 16, 128, 1, 1, 0),     # 448-pixel planes: 2 tiles of 224
    (3, 128, 14, 14, 288, 1, 1, 0),     # 9 strips on 4 waves (uneven)
    (5, 64, 28, 28, 512, 1, 1, 0),      # 20 tiles, four strips per wave
    (4, 512, 7, 7, 2048, 1, 1, 0),      # ResNet-50 layer4 expansion: 7x7 planes, 2 images per tile, the tensor's last byte patched
    (8, 512, 7, 7, 512, 1, 1, 0),       # 7x7, one channel group, two strips per wave
    (4, 256, 7, 7, 768, 1, 1, 0),       # 7x7, 4 images per tile, 3 strips per wave
    (8, 128, 7, 7, 1024, 1, 1, 0),      # 7x7, IC = 128
    (5, 512, 7, 7, 1024, 1, 1, 0),      # odd batch: not a whole number of tiles -> the ring kernel keeps it
]


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("pwr", ["1", "0"])
def test_pwr_vs_oracle(engine, pwr):
    """pwr = 1: every eligible layer on the resident-tile kernel (one tile per workgroup; the patched
    tail of the tensor in a late tile); 0: kernel disabled."""
    rng = np.random.RandomState(4242)
    env = {"QE_PWR": "0" if pwr == "0" else "2"}      # 2: every eligible layer (default 1 = whole-plane tiles only)

    def run():
        for shp in SHAPES:
            for (asgn, zeros, w_pc, bias) in [(1, False, True, True), (0, True, True, True), (1, True, False, False)]:
                case = _random_case(rng, *shp, 8, 1 if asgn else 0, 8, asgn, w_pc=w_pc, a_pc=False, zeros=zeros, bias=bias)
                y, o32, o64 = _run_case(engine, case, via_capi=True)
                assert case["path"] == 1
                _assert_conv_close(y, o64, o32, "pwr=%s %s asgn=%d zeros=%s w_pc=%s" % (pwr, shp, asgn, zeros, w_pc), case["fma"])
                if not zeros:
                    assert np.abs(y.astype(np.float64) - o64).max() <= 1e-5
    _with_env(env, run)


@pytest.mark.parametrize("groups", ["2", "4"])
def test_pwr_channel_groups(engine, groups):
    rng = np.random.RandomState(77)

    def run():
        for shp in [(2, 256, 14, 14, 1024, 1, 1, 0), (1, 128, 28, 28, 512, 1, 1, 0), (1, 64, 56, 56, 256, 1, 1, 0)]:
            case = _random_case(rng, *shp, 8, 0, 8, 1, w_pc=True, a_pc=False, zeros=True, bias=True)
            y, o32, o64 = _run_case(engine, case, via_capi=True)
            _assert_conv_close(y, o64, o32, "groups=%s %s" % (groups, shp), case["fma"])
    _with_env({"QE_PWR": "2", "QE_PWR_GROUPS": groups}, run)


def test_pwr_batch256_independence(engine, monkeypatch):
    """The batch-256 launch geometry (256 / 1024 / 3584 workgroups): image i of the batched call == the same image alone."""
    import torch
    monkeypatch.setenv("QE_PWR", "2")
    capi.reload_env()
    rng = np.random.RandomState(5)
    for (ic, hw, oc) in [(256, 14, 1024), (128, 28, 512), (64, 56, 256)]:
        n = 256
        xq = torch.from_numpy(rng.randint(-128, 128, size=(n, ic, hw, hw)).astype(np.int8)).cuda()
        wq = torch.from_numpy(rng.randint(-128, 128, size=(oc, ic, 1, 1)).astype(np.int8)).cuda()
        ws = torch.from_numpy(rng.uniform(2.5e-4, 7.5e-4, size=(oc,)).astype(np.float32)).cuda()
        wz = torch.zeros(oc, dtype=torch.float32, device="cuda")
        xs = torch.tensor([2e-3], dtype=torch.float32, device="cuda")
        xz = torch.zeros(1, dtype=torch.float32, device="cuda")
        b = torch.from_numpy(rng.normal(0, 0.1, size=(oc,)).astype(np.float32)).cuda()
        xp, xd = engine.tpack(xq, 8, True)
        wp, wd = engine.tpack(wq, 8, True)
        full = engine.quantconv2d(xp, xd, xs, xz, wp, wd, ws.view(oc, 1, 1, 1), wz.view(oc, 1, 1, 1), b, 1, 0)
        for i in (0, 1, 100, 255):
            xpi, xdi = engine.tpack(xq[i:i + 1].contiguous(), 8, True)
            one = engine.quantconv2d(xpi, xdi, xs, xz, wp, wd, ws.view(oc, 1, 1, 1), wz.view(oc, 1, 1, 1), b, 1, 0)
            assert torch.equal(full[i:i + 1], one), (ic, hw, oc, i)
        # and against plain integer arithmetic on one image
        ref = torch.nn.functional.conv2d(xq[255:256].double(), wq.double()) * (2e-3 * ws.double().view(1, oc, 1, 1)) + b.double().view(1, oc, 1, 1)
        assert (full[255:256].double() - ref).abs().max().item() <= 1e-5


WIDE_SHAPES = [
    # N, IC, H, W, OC, K, stride, pad -- deep 1x1 reductions into a multiple of 256 output channels (8-wave flat kernel)
    (2, 1024, 14, 14, 256, 1, 1, 0),    # ResNet-50 layer3 reduction: whole-plane tiles of 224 slots
    (1, 1024, 14, 14, 512, 1, 1, 0),    # layer4.0.conv1: two channel tiles
    (2, 512, 28, 28, 256, 1, 1, 0),     # layer3.0.conv1: 160-pixel tiles (5 column tiles), ragged last tile
    (1, 512, 14, 14, 1024, 1, 1, 0),    # the dense form of layer3's downsample branch
    (3, 640, 12, 12, 256, 1, 1, 0),     # 144-pixel planes, 5 stages of 128 channels
    (2, 512, 28, 28, 1024, 1, 2, 0),    # strided: gather pass + the dense problem above
]


@pytest.mark.parametrize("flat8", ["1", "0"])
def test_wide_flat_kernel_vs_oracle(engine, flat8):
    """conv_mfma_flat_kernel<8, 1, NIW, 4>: 256 output channels per workgroup (launch_conv_mfma: wide8) against the oracle and,
    bit for bit, against the 4-wave instances (same integer sums, same epilogue operation order)."""
    rng = np.random.RandomState(808)

    def run():
        for shp in WIDE_SHAPES:
            for (asgn, zeros, w_pc, bias) in [(1, False, True, True), (0, True, True, True), (1, True, False, False)]:
                case = _random_case(rng, *shp, 8, 1 if asgn else 0, 8, asgn, w_pc=w_pc, a_pc=False, zeros=zeros, bias=bias)
                y, o32, o64 = _run_case(engine, case, via_capi=True)
                assert case["path"] == 1
                _assert_conv_close(y, o64, o32, "flat8=%s %s asgn=%d zeros=%s" % (flat8, shp, asgn, zeros), case["fma"])
                if not zeros:
                    assert np.abs(y.astype(np.float64) - o64).max() <= 1e-5
                y_other = _with_env({"QE_FLAT8": "0" if flat8 == "1" else "1"}, lambda: _run_case(engine, case, via_capi=True)[0])
                assert np.array_equal(y, y_other), "wide and 4-wave flat kernels differ: %s" % (shp,)
    _with_env({"QE_FLAT8": flat8}, run)
