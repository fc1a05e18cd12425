set -o pipefail
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/r03y_conv_tests.txt 2>&1; rc=$?; tail -5 gpurun_out/r03y_conv_tests.txt; [ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r03y_bench_$i.json 2> gpurun_out/r03y_bench_$i.err || exit 1
python -c "
import json;j=json.load(open('gpurun_out/r03y_bench_$i.json'));print('bench', j['value'], j['ms_per_step'], j['roofline']['conv_stack_ms'])"
done
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --per-layer --cold --no-cpu-baseline --layers 2,6,9,29,32,35,38,41 > gpurun_out/r03y_pl.json 2> gpurun_out/r03y_pl.err; grep -E "^ *[0-9]+ (layer|conv)|sum of" gpurun_out/r03y_pl.err | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$10}' 
