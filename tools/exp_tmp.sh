cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r03zz_gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03zz_gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
rm -rf gpurun_out/r03z gpurun_out/r03z_fusedrequant
bash tools/profile_round.sh r03z > gpurun_out/r03z_profile.log 2>&1; tail -2 gpurun_out/r03z_profile.log | cut -c1-200
bash tools/profile_round.sh r03z_fusedrequant "--fused-requant" > gpurun_out/r03z_profile_fusedrequant.log 2>&1; tail -2 gpurun_out/r03z_profile_fusedrequant.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline --fused-requant > gpurun_out/r03z_bench_fusedrequant.json 2>/dev/null; tail -c 400 gpurun_out/r03z_bench_fusedrequant.json; echo
timeout -k 10 300 python bench.py > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err; tail -c 300 gpurun_out/r03z_bench.json
