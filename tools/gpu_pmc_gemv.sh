#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc_gemv_sq -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-prefill 2>&1 | tail -2 ) > $O/pmc_gemv_sq.log
( timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $O/pmc_gemv_sq2 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-prefill 2>&1 | tail -2 ) > $O/pmc_gemv_sq2.log
python tools/pmc_summary.py $O $O/pmc_gemv_summary.txt gemv_cdna4
find $O -name "*.db" -delete
tail -n 2 $O/pmc_gemv_sq.log
