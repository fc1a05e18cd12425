#!/usr/bin/env python3
"""Per-phase cycle breakdown of linear_mfma_kernel from the -DQE_STAMP diagnostic build.
usage: QE_LIB=quantize_amd/_ext/libqe_hip_stamp.so python tools/stamp_linear.py B K O [B K O ...]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from quantize_amd import capi
L = capi.lib()
dev = torch.device("cuda", 0)
names = ["prologue", "wait DMA", "barrier", "DMA issue", "frag reads + MFMA", "-", "epilogue issue", "store drain"]
v = [int(a) for a in sys.argv[1:]]
for B, K, O in zip(v[0::3], v[1::3], v[2::3]):
    x = torch.randint(0, 256, (B * K,), device=dev, dtype=torch.int32).to(torch.uint8)
    w = torch.randint(0, 256, (O * K,), device=dev, dtype=torch.int32).to(torch.uint8)
    one, zero = torch.full((1,), 2e-3, device=dev), torch.zeros(1, device=dev)
    xq, wq = capi.qparam(x, 8, 1, one, zero), capi.qparam(w, 8, 1, one, zero)
    out = torch.empty((B, O), device=dev)
    buf = torch.zeros((1 << 20) * 40, dtype=torch.int64, device=dev)
    L.qe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    capi.quantlinear(xq, wq, None, B, K, O, out=out); torch.cuda.synchronize()
    buf.zero_(); capi.quantlinear(xq, wq, None, B, K, O, out=out); torch.cuda.synchronize()
    d = buf.view(-1, 10); d = d[d[:, 8] > 0].double()
    tot = d[:, 8].mean().item()
    print("linear %dx%d->%d: %d waves" % (B, K, O, d.shape[0]))
    for i in range(8):
        m = d[:, i].mean().item()
        print("   %-20s %9.0f cyc  %5.1f%%" % (names[i], m, 100 * m / tot))
    print("   %-20s %9.0f cyc; span of starts %.0f cyc" % ("total per wave", tot, (d[:, 9].max() - d[:, 9].min()).item()))
    L.qe_debug_set_stamp_buffer(None)
