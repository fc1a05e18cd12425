AB_EXTRA="--cold --layers 1,5,8" bash tools/ab_env.sh QE_FLAT_W448 0 1 0 1 2>&1 | tail -3
grep -E "^ *[0-9]+ layer" gpurun_out/ab_QE_FLAT_W448_0.err gpurun_out/ab_QE_FLAT_W448_1.err | awk '{print $1,$2,$3,$11}'
QE_FLAT_W448=1 timeout -k 10 300 python -m pytest tests/test_conv_gpu.py -x -q -k "headline or family" 2>&1 | tail -2
