set -o pipefail
timeout -k 10 600 python -m pytest tests/test_linear_gpu.py -x -q > gpurun_out/r03x_lin_test.txt 2>&1; rc=$?; tail -15 gpurun_out/r03x_lin_test.txt; [ $rc -eq 0 ] || exit 1
for v in 0 1 2 auto; do
if [ $v = auto ]; then unset QE_LIN8; else export QE_LIN8=$v; fi
timeout -k 10 200 python tools/bench_linear.py --steps 5 > gpurun_out/r03x_linear_$v.json; python -c "
import json;j=json.load(open('gpurun_out/r03x_linear_$v.json'));print('QE_LIN8=$v', j['value'], j['ms_per_step'], j.get('per_shape') or j)"
done
export QE_LIN8=1
QE_LIB=quantize_amd/_ext/libqe_hip_stamp.so timeout -k 10 120 python tools/stamp_linear.py 50432 768 768 50432 768 3072 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03x_stamp_linear8c.txt
