#!/bin/bash
# Memory-path PMC passes over the whole conv stack (separate rocprofv3 --pmc runs, no tracing); per-kernel means by tools/pmc_summary.py
# usage: tools/pmc_stack.sh outdir
set -u
OUT=${1:-gpurun_out/pmcstack}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run a GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum &&
run c TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run h TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $OUT
