#!/usr/bin/env python3
"""quantlinear_float_input (fp32 activations x packed 8-bit weights) at the ViT-B/16 shapes: the bf16 x 3 MFMA kernel against the
order-preserving fp32 kernel (QE_LIN_F32_MFMA=0).  usage: python tools/bench_linear_f32.py"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from quantize_amd import capi
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
res = {}
for (B, K, O) in [(50432, 768, 768), (50432, 768, 3072), (50432, 3072, 768)]:
    x = torch.randn(B, K, generator=g, device=dev)
    w = torch.randint(0, 256, (O * K,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    sw = torch.rand(O, generator=g, device=dev) * 5e-4 + 2.5e-4; zw = torch.zeros(O, device=dev)
    wq = capi.qparam(w, 8, 1, sw, zw)
    out = torch.empty(B, O, device=dev)
    row = {}
    for mode in ("1", "0"):
        os.environ["QE_LIN_F32_MFMA"] = mode; capi.reload_env()
        path = capi.linear_float_input_path(x, wq, B, K, O)
        capi.quantlinear_float_input(x, wq, None, O, out=out); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5 if mode == "1" else 2
        a.record()
        for _ in range(n): capi.quantlinear_float_input(x, wq, None, O, out=out)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        row["mfma" if path == 1 else "fp32_chain"] = {"ms": round(ms, 4), "TFLOPs_fp32_equiv": round(2.0 * B * K * O / ms / 1e9, 1),
                                                      "GBs": round((4 * B * K + O * K + 4 * B * O) / ms / 1e6, 1)}
    res["%dx%d->%d" % (B, K, O)] = row
print(json.dumps(res))
