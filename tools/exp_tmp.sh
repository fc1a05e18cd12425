timeout -k 10 600 python -m pytest tests/test_linear_gpu.py tests/test_packed_modules_gpu.py -x -q 2>&1 | tail -3
for im in 256 64 16; do
timeout -k 10 200 python tools/bench_linear.py --steps 5 --images $im 2>/dev/null > gpurun_out/r03zz_lin_${im}_auto.json; python -c "
import json;j=json.load(open('gpurun_out/r03zz_lin_${im}_auto.json'));print('images=$im auto', round(j['value']), round(j['ms_per_step'],3), {k:v['ms'] for k,v in j['per_shape'].items()})"
done | tee gpurun_out/r03zz_lin_auto.txt
