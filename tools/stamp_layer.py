#!/usr/bin/env python3
"""Per-phase cycle breakdown of conv_mfma_kernel from the -DQE_STAMP diagnostic build.
usage: QE_LIB=quantize_amd/_ext/libqe_hip_stamp.so python tools/stamp_layer.py 28 3 13 ..."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from quantize_amd import capi, resnet50
import argparse
from bench import Layer

class A: pass
args = A(); args.a_bits = 8; args.w_bits = 8; args.asymmetric = False; args.per_call_prepare = True; args.float_input = False
L = capi.lib()
dev = torch.device("cuda", 0)
specs = resnet50.conv_layers()
names = ["A issue / prologue", "X wait+transpose+LDSwr", "barrier1", "X(s+1) issue", "MFMA phase", "barrier2", "pre-epilogue | flat: epilogue", "epilogue | flat: store drain", "total"]
for idx in [int(v) for v in sys.argv[1:]]:
    layer = Layer(idx, specs[idx], 256, dev, args, 0, capi, resnet50, torch)
    nblocks_max = 1 << 20
    buf = torch.zeros(nblocks_max * 4 * 10, dtype=torch.int64, device=dev)
    L.qe_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    layer.run(st); torch.cuda.synchronize()
    buf.zero_()
    if os.environ.get("QE_STAMP_COLD"):   # inputs from HBM, as inside the stack
        torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
    layer.run(st); torch.cuda.synchronize()
    dall = buf.view(-1, 10)
    groups = [("all waves", dall)]
    if os.environ.get("QE_STAMP_WS"):   # warp-specialised kernel: 8 waves per block, 0-3 consumers, 4-7 producers
        d8 = dall.view(-1, 8, 10)
        groups = [("consumers", d8[:, :4].reshape(-1, 10)), ("producers", d8[:, 4:].reshape(-1, 10))]
    for gname, d in groups:
        d = d[d[:, 8] > 0].double()
        print("layer %d %s %s [%s]: %d waves" % (idx, specs[idx].name, tuple(specs[idx][1:]), gname, d.shape[0]))
        tot = d[:, 8].mean().item()
        for i in range(8):
            m = d[:, i].mean().item()
            print("   %-32s %9.0f cyc  %5.1f%%" % (names[i], m, 100 * m / tot))
        print("   %-32s %9.0f cyc; span first->last start %.0f" % ("total per wave", tot, (d[:, 9].max() - d[:, 9].min()).item()))
    L.qe_debug_set_stamp_buffer(None)
