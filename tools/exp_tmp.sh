cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r03z
bash tools/profile_round.sh r03z > gpurun_out/r03z_profile.log 2>&1; tail -2 gpurun_out/r03z_profile.log | cut -c1-400
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --per-layer --cold --no-cpu-baseline > gpurun_out/r03z_pl.json 2> gpurun_out/r03z_pl.err; cp gpurun_out/per_layer.json gpurun_out/r03z_per_layer_cold.json
python tools/stack_timeline.py gpurun_out/r03z/trace gpurun_out/r03z_per_layer_cold.json > gpurun_out/r03z_stack_timeline.txt 2>&1; tail -1 gpurun_out/r03z_stack_timeline.txt
timeout -k 10 300 python bench.py > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err; tail -c 900 gpurun_out/r03z_bench.json
