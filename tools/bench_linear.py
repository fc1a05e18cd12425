#!/usr/bin/env python3
"""Secondary measurement (BASELINE config 5): the packed W8A8 linear layers of ViT-B/16 at batch 256
(50,432 token rows) through qe_quantlinear.  One step = the 73 linears of one forward (12 blocks x {q, k, v, proj:
768->768; fc1: 768->3072; fc2: 3072->768} + head 768->1000 on the 256 class tokens), as independent packed problems
resident in HBM.  Prints ONE JSON line; `roofline` prices the step against HBM (algorithmic bytes = packed x once +
packed w once + fp32 out once per layer) and states the int8 MFMA fraction next to it.
usage: python tools/bench_linear.py [--steps 10] [--warmup 2] [--images 256]"""
import argparse, ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from quantize_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--images", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda", 0)
L = capi.lib()
tokens = args.images * 197
shapes = []
for blk in range(12):
    shapes += [("blk%d.%s" % (blk, n), tokens, 768, 768) for n in ("q", "k", "v", "proj")]
    shapes += [("blk%d.fc1" % blk, tokens, 768, 3072), ("blk%d.fc2" % blk, tokens, 3072, 768)]
shapes.append(("head", args.images, 768, 1000))
g = torch.Generator(device=dev); g.manual_seed(7)
# operands are shared per distinct (B, K) / (O, K) to fit comfortably in HBM; outputs are per shape class
acts, wts, outs, calls = {}, {}, {}, []
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
for name, B, K, O in shapes:
    if (B, K) not in acts:
        x = torch.randint(0, 256, (B * K,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
        sx = torch.full((1,), 2e-3, device=dev); zx = torch.zeros(1, device=dev)
        acts[(B, K)] = capi.qparam(x, 8, 1, sx, zx)
    if (O, K) not in wts:
        w = torch.randint(0, 256, (O * K,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
        sw = torch.rand(O, generator=g, device=dev) * 5e-4 + 2.5e-4; zw = torch.zeros(O, device=dev)
        wts[(O, K)] = (capi.qparam(w, 8, 1, sw, zw), torch.randn(O, generator=g, device=dev) * 0.1)
    if (B, O) not in outs:
        outs[(B, O)] = torch.empty((B, O), dtype=torch.float32, device=dev)
    xq, (wq, bias), out = acts[(B, K)], wts[(O, K)], outs[(B, O)]
    assert capi.linear_path(xq, wq, B, K, O) == 1
    calls.append((ctypes.byref(xq), ctypes.byref(wq), bias.data_ptr(), B, K, O, out.data_ptr()))
bytes_step = sum(B * K + O * K + 4 * B * O for _, B, K, O in shapes)
ops_step = sum(2 * B * K * O for _, B, K, O in shapes)

def step():
    for c in calls:
        rc = L.qe_quantlinear(*c, sp)
        assert rc == 0, rc
with torch.cuda.stream(stream):
    for _ in range(args.warmup): step()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.steps): step()
    e1.record(stream)
    stream.synchronize()
ms = e0.elapsed_time(e1) / args.steps
per = {}
with torch.cuda.stream(stream):
    for (name, B, K, O), c in zip(shapes, calls):
        if (B, K, O) in per: continue
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.qe_quantlinear(*c, sp); a.record(stream)
        for _ in range(5): L.qe_quantlinear(*c, sp)
        b.record(stream); stream.synchronize()
        t = a.elapsed_time(b) / 5
        per["%dx%d->%d" % (B, K, O)] = {"ms": round(t, 4), "GBs": round((B * K + O * K + 4 * B * O) / t / 1e6, 1),
                                         "TOPs": round(2 * B * K * O / t / 1e9, 1)}
print(json.dumps({
    "metric": "quant-linear images/sec at batch %d (ViT-B/16 linear layers, W8A8)" % args.images,
    "value": args.images / (ms * 1e-3), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
    "ms_per_step": ms, "higher_is_better": True, "dtype": "int8", "data": "synthetic",
    "config": {"workload": "ViT-B/16 W8A8: 73 packed linears per forward, %d token rows, fp32 outputs" % tokens},
    "roofline": {"bound": "hbm", "kernel": "linear_mfma_kernel", "achieved": bytes_step / ms / 1e6, "peak": 8000.0,
                 "unit": "GB/s", "frac": bytes_step / ms / 1e6 / 8000.0, "traffic": None,
                 "bytes_per_launch": bytes_step / len(shapes), "avg_launch_ms": ms / len(shapes),
                 "int8_tops": ops_step / ms / 1e9, "mfma_frac": ops_step / ms / 1e9 / 5000.0},
    "per_shape": per}))
