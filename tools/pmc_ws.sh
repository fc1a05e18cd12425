#!/bin/bash
OUT=gpurun_out/pmc_ws; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for np in 0 1; do
export QE_WS_NOPAD=$np
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a$np -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers 29 > $OUT/a$np.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/b$np -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers 29 > $OUT/b$np.log 2>&1
echo "== NOPAD=$np"; python tools/pmc_summary.py $OUT/a$np | grep -A12 "ws_kernel"; python tools/pmc_summary.py $OUT/b$np | grep -A12 "ws_kernel"
done
