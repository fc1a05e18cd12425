set -o pipefail
timeout -k 10 600 python -m pytest tests/test_requant_gpu.py tests/test_packed_modules_gpu.py -x -q > gpurun_out/r03y_rq_test.txt 2>&1; rc=$?; tail -5 gpurun_out/r03y_rq_test.txt; [ $rc -eq 0 ] || exit 1
for v in 0 1 0 1; do
  QE_RQ_PATCH=$v timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --fused-requant > gpurun_out/r03y_fused_$v.json 2> gpurun_out/r03y_fused_$v.err || exit 1
  python -c "
import json;j=json.load(open('gpurun_out/r03y_fused_$v.json'));print('QE_RQ_PATCH=$v', j['value'], j['ms_per_step'], j['roofline']['conv_stack_ms'])"
done | tee gpurun_out/r03y_ab_rq_patch.txt
QE_RQ_PATCH=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --per-layer --cold --no-cpu-baseline --fused-requant --layers 0,2,6,12,16,19,25,29,32 > gpurun_out/r03y_fused_pl.json 2> gpurun_out/r03y_fused_pl.err; grep -E "^ *[0-9]+ (layer|conv)|sum of" gpurun_out/r03y_fused_pl.err | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$10}'
