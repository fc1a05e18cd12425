timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02z_gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02z_gputest.txt
timeout -k 10 300 python bench.py > gpurun_out/r02z_bench.json 2> gpurun_out/r02z_bench.err; tail -c 1200 gpurun_out/r02z_bench.json
bash tools/profile_round.sh r02z > gpurun_out/r02z_profile.log 2>&1; tail -3 gpurun_out/r02z_profile.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --per-layer --cold --no-cpu-baseline > gpurun_out/r02z_pl.json 2> gpurun_out/r02z_pl.err; cp gpurun_out/per_layer.json gpurun_out/r02z_per_layer_cold.json
python tools/stack_timeline.py gpurun_out/r02z/trace gpurun_out/r02z_per_layer_cold.json > gpurun_out/r02z_stack_timeline.txt 2>&1; tail -2 gpurun_out/r02z_stack_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02z_tpack -- python3 tools/bench_tpack.py > gpurun_out/r02z_bench_tpack.json 2> gpurun_out/r02z_tpack.err; tail -c 300 gpurun_out/r02z_bench_tpack.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02z_linear -- python3 tools/bench_linear.py > gpurun_out/r02z_bench_linear.json 2> gpurun_out/r02z_linear.err; tail -c 300 gpurun_out/r02z_bench_linear.json
