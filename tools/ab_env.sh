#!/bin/bash
# A/B an environment knob over the per-layer isolated times: tools/ab_env.sh VAR v1 v2 ...
# AB_EXTRA=--cold times every call after a 1 GiB overwrite (inputs from HBM, as inside the stack)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --per-layer $AB_EXTRA --no-cpu-baseline > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.err
done
python - "$VAR" "$@" <<'PY'
import re, sys, json
var=sys.argv[1]; vals=sys.argv[2:]
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*(\d+) (\S+)\s+(\[.*?\])\s+\S+\s+([\d.]+) ms",l)
        if m: d[int(m.group(1))]=(m.group(2),m.group(3),float(m.group(4)))
    return d
D=[load('gpurun_out/ab_%s_%s.err'%(var,v)) for v in vals]
print('layer', ' '.join('%s=%s'%(var,v) for v in vals))
for i in sorted(D[0]):
    t=[d[i][2] for d in D]
    if (max(t)-min(t))/min(t) > 0.03: print(i, D[0][i][0], D[0][i][1], ' '.join('%.4f'%x for x in t))
print('sum', ' '.join('%.3f'%sum(d[i][2] for i in d) for d in D))
for v in vals:
    j=json.load(open('gpurun_out/ab_%s_%s.json'%(var,v))); print(v, 'img/s %.0f'%j['value'])
PY
