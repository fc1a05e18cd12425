#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the bench command, (2) separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE (HBM traffic), as MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots" prescribe.
# usage: tools/profile_round.sh r01b ["--float-input" | "--fused-requant" | "--w-bits 4 --a-bits 4"]
#        (with extra bench arguments the summary goes to profiles/<tag>_* only, never to profiles/traffic.json)
TAG=${1:-r01}
EXTRA=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo trace failed
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || echo fetch failed
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || echo write failed
python tools/profile_summary.py $OUT $TAG "$EXTRA"
