timeout -k 10 600 python -m pytest tests/test_conv_f32_gpu.py -x -q -k non_finite 2>&1 | tail -8
