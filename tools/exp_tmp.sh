set -o pipefail
for r in 1 2 3; do for v in 1 2; do
  QE_PWR=$v timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r03t_pwr_$v.json 2> gpurun_out/r03t_pwr_$v.err || exit 1
  python -c "
import json;j=json.load(open('gpurun_out/r03t_pwr_$v.json'));print('QE_PWR=$v', j['value'], j['ms_per_step'], j['roofline']['conv_stack_ms'])"
done; done | tee gpurun_out/r03t_ab_pwr_stack.txt
