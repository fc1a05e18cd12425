cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "--float-input" "--w-bits 4 --a-bits 4" "--asymmetric"; do
  S=$(echo "$V" | tr -d ' -' ); bash tools/profile_round.sh r03z_$S "$V" > gpurun_out/r03z_profile_$S.log 2>&1; tail -1 gpurun_out/r03z_profile_$S.log | cut -c1-120
  timeout -k 10 200 python bench.py --no-cpu-baseline $V > gpurun_out/r03z_bench_$S.json 2> gpurun_out/r03z_bench_$S.err; python -c "
import json;j=json.load(open('gpurun_out/r03z_bench_$S.json'));print('$S', round(j['value']), round(j['roofline']['conv_stack_ms'],3), round(j['roofline']['frac'],3), j['roofline']['traffic'])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03z_linear -- python3 tools/bench_linear.py > gpurun_out/r03z_bench_linear.json 2> gpurun_out/r03z_linear.err; tail -c 200 gpurun_out/r03z_bench_linear.json
