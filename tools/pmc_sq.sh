cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcsq
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcsq/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcsq/a.log 2>&1 || echo failed a
timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --output-format csv -d gpurun_out/pmcsq/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcsq/b.log 2>&1 || echo failed b
ls gpurun_out/pmcsq/*/runc/ | head
