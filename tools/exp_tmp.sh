cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r03zz_gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03zz_gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
