#!/bin/bash
# PMC passes (separate from --kernel-trace/--stats runs, as the HBM/rocprofv3 guide prescribes) for a few layers.
# usage: tools/pmc_layers.sh "3,13,28,29" outdir
set -u
LAYERS=${1:-"3,13,28,29"}
OUT=${2:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name counters...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers $LAYERS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES &&
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM &&
run fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
find $OUT -name "*counter_collection.csv" | head
