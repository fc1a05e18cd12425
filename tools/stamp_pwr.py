#!/usr/bin/env python3
"""Per-phase cycle breakdown of conv_pwr_kernel (one tile per workgroup) from the -DQE_STAMP diagnostic build.
usage: QE_LIB=quantize_amd/_ext/libqe_hip_stamp.so python tools/stamp_pwr.py [--batch N] 26 13 3"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from quantize_amd import capi, resnet50
from bench import Layer

class A: pass
args = A(); args.a_bits = 8; args.w_bits = 8; args.asymmetric = False; args.per_call_prepare = True; args.float_input = False
args.fused_requant = False
L = capi.lib()
dev = torch.device("cuda", 0)
specs = resnet50.conv_layers()
argv = sys.argv[1:]
batch = 256
if argv and argv[0] == "--batch":
    batch = int(argv[1]); argv = argv[2:]
names = ["prologue (requests issued)", "tile landed + recode", "barrier", "K loops", "epilogues (issue)", "weight waits", "store drain", "-", "total"]
for idx in [int(v) for v in argv]:
    layer = Layer(idx, specs[idx], batch, dev, args, 0, capi, resnet50, torch)
    buf = torch.zeros((1 << 20) * 8 * 10, dtype=torch.int64, device=dev)
    L.qe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    layer.run(st); torch.cuda.synchronize()
    buf.zero_()
    if os.environ.get("QE_STAMP_COLD"):
        torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
    layer.run(st); torch.cuda.synchronize()
    d = buf.view(-1, 10)
    d = d[d[:, 8] > 0].double()
    print("layer %d %s %s batch %d: %d waves" % (idx, specs[idx].name, tuple(specs[idx][1:]), batch, d.shape[0]))
    tot = d[:, 8].mean().item()
    for i in range(7):
        m = d[:, i].mean().item()
        print("   %-32s %9.0f cyc  %5.1f%%" % (names[i], m, 100 * m / tot))
    t0 = d[:, 9].min().item()
    ends = d[:, 9] + d[:, 8]
    print("   %-32s %9.0f cyc; first start -> last start %.0f, first start -> last end %.0f, first end %.0f (100 MHz ticks? no: shader clocks)" % (
        "total per wave", tot, (d[:, 9].max() - t0).item(), (ends.max() - t0).item(), (ends.min() - t0).item()))
    L.qe_debug_set_stamp_buffer(None)
