#!/usr/bin/env python3
"""tpack / tunpack throughput on the device (algorithmic bytes: sizeof(T) + b/8 per element for pack, b/8 + 1 for
unpack) against the HBM roofline.  usage: python tools/bench_tpack.py"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import quantize_amd.engine as engine
dev = torch.device("cuda", 0)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
rows = []
for shape, name in (((256, 3, 224, 224), "input batch"), ((256, 256, 56, 56), "layer1 activation"), ((2048, 512, 3, 3), "layer4 3x3 weight")):
    n = 1
    for v in shape: n *= v
    for dtype, esz in ((torch.float32, 4), (torch.int8, 1)):
        for bits in (8, 4, 2):
            lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
            x = torch.randint(lo, hi + 1, shape, device=dev, dtype=torch.int32).to(dtype)
            ms = t(lambda: engine.tpack(x, bits, True))
            p, d = engine.tpack(x, bits, True)
            ms_u = t(lambda: engine.tunpack(p, d))
            rows.append({"tensor": name, "elements": n, "dtype": str(dtype).replace("torch.", ""), "bits": bits,
                         "tpack_ms": round(ms, 4), "tpack_GBs": round(n * (esz + bits / 8) / ms / 1e6, 1),
                         "tunpack_ms": round(ms_u, 4), "tunpack_GBs": round(n * (bits / 8 + 1) / ms_u / 1e6, 1)})
            print(rows[-1], file=sys.stderr)
print(json.dumps({"metric": "tpack/tunpack throughput", "unit": "GB/s algorithmic", "peak": 8000.0, "rows": rows}))
