#!/bin/bash
# Isolated (cold) time of some layers at several batch sizes: the intercept of t(batch) is what a launch costs besides
# its bytes (ramp-up, tail, boundary), the slope the marginal rate.  tools/batch_scaling.sh "3,13,26,28,41" 64 128 256 512
LAYERS=$1; shift
for b in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --per-layer --cold --no-cpu-baseline --batch $b --layers $LAYERS > gpurun_out/bs_$b.json 2> gpurun_out/bs_$b.err
done
python - "$@" <<'PY'
import re, sys
vals=sys.argv[1:]
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*(\d+) (\S+)\s+(\[.*?\])\s+\S+\s+([\d.]+) ms",l)
        if m: d[int(m.group(1))]=(m.group(2),m.group(3),float(m.group(4)))
    return d
D=[load('gpurun_out/bs_%s.err'%v) for v in vals]
print('layer', ' '.join('N=%s'%v for v in vals), ' intercept_us  marginal_us_per_256')
for i in sorted(D[0]):
    t=[d[i][2]*1e3 for d in D]
    n=[float(v) for v in vals]
    # least squares t = a + b n
    mn=sum(n)/len(n); mt=sum(t)/len(t)
    b=sum((x-mn)*(y-mt) for x,y in zip(n,t))/sum((x-mn)**2 for x in n)
    a=mt-b*mn
    print(i, D[0][i][0], D[0][i][1], ' '.join('%.1f'%x for x in t), ' a=%.1f  b256=%.1f'%(a,b*256))
PY
