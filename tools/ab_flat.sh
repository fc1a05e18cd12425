#!/bin/bash
# A/B the flat kernel's tile width on the 1x1 layers (isolated per-layer times).
for v in 7 4; do
  QE_FLAT_NIW=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --per-layer --no-cpu-baseline > gpurun_out/ab_niw$v.json 2> gpurun_out/ab_niw$v.err
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*(\d+) (\S+)\s+(\[.*?\])\s+\S+\s+([\d.]+) ms",l)
        if m: d[int(m.group(1))]=(m.group(2),m.group(3),float(m.group(4)))
    return d
a=load('gpurun_out/ab_niw7.err'); b=load('gpurun_out/ab_niw4.err')
for i in sorted(a):
    if abs(a[i][2]-b[i][2])/a[i][2] > 0.03: print(i,a[i][0],a[i][1],'niw7 %.4f niw4 %.4f'%(a[i][2],b[i][2]))
PY
