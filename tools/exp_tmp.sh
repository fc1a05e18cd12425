timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_fuzz_gpu.py tests/test_prepared_gpu.py -x -q > gpurun_out/r03zz_stem_test.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03zz_stem_test.txt
AB_EXTRA="--cold --layers 0" bash tools/ab_env.sh QE_STEM_SWAP 0 1 0 1 2>&1 | tail -4
grep -E "^ *0 conv1" gpurun_out/ab_QE_STEM_SWAP_0.err gpurun_out/ab_QE_STEM_SWAP_1.err
for v in 0 1 0 1; do QE_STEM_SWAP=$v timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r03zz_stem_$v.json; python -c "
import json;j=json.load(open('gpurun_out/r03zz_stem_$v.json'));print('QE_STEM_SWAP=$v', round(j['value']), round(j['roofline']['conv_stack_ms'],4))"; done | tee gpurun_out/r03zz_ab_stem_swap.txt
