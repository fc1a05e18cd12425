set -o pipefail
timeout -k 10 600 python -m pytest tests/test_linear_gpu.py -x -q > gpurun_out/r03z_lin_test.txt 2>&1; tail -4 gpurun_out/r03z_lin_test.txt
bash tools/final_round.sh r03z 2 2>&1 | tail -40
for v in "" "--branch-streams"; do
  timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline $v > gpurun_out/r03z_bench_br$( [ -n "$v" ] && echo 1 || echo 0).json 2> /dev/null
done
python -c "
import json
for k in (0,1):
    j=json.load(open('gpurun_out/r03z_bench_br%d.json'%k)); print('branch-streams', k, j['value'], j['ms_per_step'], j['roofline']['conv_stack_ms'])" | tee gpurun_out/r03z_ab_branch_streams.txt
bash tools/pmc_sq.sh > gpurun_out/r03z_pmc_sq.log 2>&1; python tools/pmc_summary.py gpurun_out/pmcsq > gpurun_out/r03z_pmc_mfma.txt 2>&1; tail -30 gpurun_out/r03z_pmc_mfma.txt
