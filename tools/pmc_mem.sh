#!/bin/bash
# Memory-path PMC passes for a few layers (separate rocprofv3 --pmc runs, no tracing): where do the write-heavy / read-heavy
# 1x1 layers wait?  usage: tools/pmc_mem.sh "26,28,3" outdir
set -u
LAYERS=${1:-"26,28,3"}
OUT=${2:-gpurun_out/pmcmem}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers $LAYERS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run a GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum &&
run b TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr TCC_TAG_STALL_sum &&
run c TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run d SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD &&
run e TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_sum
python3 tools/pmc_summary.py $OUT
