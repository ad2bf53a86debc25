#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_gemm_sq -o pmc -- tools/ubench/gemm_ubench 103 2>&1 | tail -3 ) > $O/pmc_gemm_sq.log
( timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD -d $O/pmc_gemm_sq2 -o pmc -- tools/ubench/gemm_ubench 103 2>&1 | tail -3 ) > $O/pmc_gemm_sq2.log

tail -2 $O/pmc_gemm_sq.log $O/pmc_gemm_sq2.log
python tools/pmc_summary.py $O $O/pmc_gemm_summary.txt gemm_cdna4
find $O -name "*.db" -delete
