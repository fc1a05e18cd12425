#!/bin/bash
# Round-end measurements.  usage: tools/final_round.sh <tag> [part]   part 1: tests, bench, headline profile, per-layer;
# part 2: the variants (float-input, fused re-quantisation, W4A4, asymmetric), packing and linear.
TAG=${1:-r03z}; PART=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$PART" = "1" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_gputest.txt
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1; tail -3 gpurun_out/${TAG}_profile.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --per-layer --cold --no-cpu-baseline > gpurun_out/${TAG}_pl.json 2> gpurun_out/${TAG}_pl.err; cp gpurun_out/per_layer.json gpurun_out/${TAG}_per_layer_cold.json
python tools/stack_timeline.py gpurun_out/${TAG}/trace gpurun_out/${TAG}_per_layer_cold.json > gpurun_out/${TAG}_stack_timeline.txt 2>&1; tail -2 gpurun_out/${TAG}_stack_timeline.txt
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench2.json 2> gpurun_out/${TAG}_bench2.err; tail -c 400 gpurun_out/${TAG}_bench2.json
else
for V in "--float-input" "--fused-requant" "--w-bits 4 --a-bits 4" "--asymmetric"; do
  S=$(echo "$V" | tr -d ' -' ); bash tools/profile_round.sh ${TAG}_$S "$V" > gpurun_out/${TAG}_profile_$S.log 2>&1; tail -2 gpurun_out/${TAG}_profile_$S.log
  timeout -k 10 200 python bench.py --no-cpu-baseline $V > gpurun_out/${TAG}_bench_$S.json 2> gpurun_out/${TAG}_bench_$S.err; tail -c 600 gpurun_out/${TAG}_bench_$S.json; echo
done
timeout -k 10 200 python bench.py --no-cpu-baseline --fused-requant --two-pass > gpurun_out/${TAG}_bench_twopass.json 2> gpurun_out/${TAG}_bench_twopass.err; tail -c 300 gpurun_out/${TAG}_bench_twopass.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_tpack -- python3 tools/bench_tpack.py > gpurun_out/${TAG}_bench_tpack.json 2> gpurun_out/${TAG}_tpack.err; tail -c 300 gpurun_out/${TAG}_bench_tpack.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_linear -- python3 tools/bench_linear.py > gpurun_out/${TAG}_bench_linear.json 2> gpurun_out/${TAG}_linear.err; tail -c 300 gpurun_out/${TAG}_bench_linear.json
fi
