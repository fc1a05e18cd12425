timeout -k 10 300 python tools/bench_linear_f32.py 2>/dev/null | tee gpurun_out/r03z_bench_linear_f32.json
