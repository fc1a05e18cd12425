cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r03z
bash tools/profile_round.sh r03z > gpurun_out/r03z_profile.log 2>&1; tail -2 gpurun_out/r03z_profile.log | cut -c1-300
timeout -k 10 300 python bench.py > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err; tail -c 500 gpurun_out/r03z_bench.json
