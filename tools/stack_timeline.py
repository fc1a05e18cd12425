#!/usr/bin/env python3
"""In-stack view of one bench step from a rocprofv3 --kernel-trace run: every kernel's duration, the idle gap
before it, and the isolated (--per-layer) time of the same layer next to it.
usage: python tools/stack_timeline.py gpurun_out/<tag>/trace [gpurun_out/per_layer.json]"""
import csv, glob, json, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_conv = lambda n: any(k in n for k in ("conv_mfma", "conv_flatd", "conv_pwr", "conv_c3")) and "prep" not in n
main = [i for i, r in enumerate(rows) if is_conv(r["Kernel_Name"])]
assert len(main) % 53 == 0, len(main)
first = main[-53]
# include a prep kernel that belongs to the first layer of the step
while first > 0 and ("prep" in rows[first - 1]["Kernel_Name"] or "subsample" in rows[first - 1]["Kernel_Name"]
                     or "tunpack" in rows[first - 1]["Kernel_Name"]):
    first -= 1
last = main[-1]
iso = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
t_prev = None
layer = -1
tot_main = tot_prep = tot_gap = 0.0
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void qe::", "")[:60]
    gap = (s - t_prev) / 1e3 if t_prev is not None else 0.0
    dur = (e - s) / 1e3
    is_prep = not is_conv(name)   # weight prep, strided gather, code expansion
    if not is_prep:
        layer += 1
        tot_main += dur
    else:
        tot_prep += dur
    tot_gap += max(gap, 0.0)
    extra = ""
    if iso and not is_prep:
        extra = "  isolated %.1f us  %s" % (iso[layer]["ms"] * 1e3, iso[layer]["shape"])
    print("%-5s gap %6.1f us  dur %7.1f us  %s%s" % ("aux" if is_prep else "L%d" % layer, gap, dur, short, extra))
    t_prev = e
print("main %.3f ms, aux (prep/gather) %.3f ms, gaps %.3f ms, span %.3f ms" % (
    tot_main / 1e3, tot_prep / 1e3, tot_gap / 1e3,
    (int(rows[last]["End_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e6))
