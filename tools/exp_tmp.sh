set -o pipefail
timeout -k 10 600 python -m pytest tests/test_requant_gpu.py tests/test_packed_modules_gpu.py -x -q > gpurun_out/r03u_rq_test.txt 2>&1; rc=$?; tail -5 gpurun_out/r03u_rq_test.txt; [ $rc -eq 0 ] || exit 1
AB_EXTRA="--cold --fused-requant --layers 3,5,13,14,15,26,28,30" bash tools/ab_env.sh QE_PWR_RQ 0 1 > gpurun_out/r03u_ab_pwr_rq_layers.txt 2>&1
grep -E "^ *[0-9]+ layer" gpurun_out/ab_QE_PWR_RQ_0.err gpurun_out/ab_QE_PWR_RQ_1.err
for v in 0 1 0 1; do
  QE_PWR_RQ=$v timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --fused-requant > gpurun_out/r03u_fused_$v.json 2> gpurun_out/r03u_fused_$v.err || exit 1
  python -c "
import json;j=json.load(open('gpurun_out/r03u_fused_$v.json'));print('QE_PWR_RQ=$v', j['value'], j['ms_per_step'], j['roofline']['conv_stack_ms'])"
done | tee gpurun_out/r03u_ab_pwr_rq.txt
