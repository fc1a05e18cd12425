#!/usr/bin/env python3
"""Summarise tools/pmc_layers.sh output: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_mfma" not in k and "conv_flatd" not in k and "conv_pwr" not in k:
            continue
        key = (k[:70], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print(key)
    for c, v in sorted(acc[key].items()):
        print("    %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
