#!/bin/bash
# prefill GEMM: timing of the product build, SQ counters (LDS conflicts, matrix-pipe busy), then the timing-only probes of an AWQ_PROBES=1 build
# usage: gpu_gemm_probe.sh <tag> [probe-lib.so]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp AWQ_TUNING=1
LIB=llm_awq_amd/lib/libawq_cdna4.so
( timeout 300 tools/ubench/gemm_ubench 103 1000103 2>&1 ) > $O/gemm_product.txt
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_gemm_sq -o pmc -- tools/ubench/gemm_ubench 1000103 2>&1 | tail -3 ) > $O/pmc_gemm_sq.log
python tools/pmc_summary.py $O $O/pmc_gemm_summary.txt gemm_cdna4 > /dev/null 2>&1
find $O -name "*.db" -delete
if [ -n "${2:-}" ]; then
  cp $LIB /tmp/lib_product.so; cp $2 $LIB
  ( timeout 400 tools/ubench/gemm_ubench 103 9103 10103 200103 400103 2>&1 ) > $O/gemm_probes.txt
  cp /tmp/lib_product.so $LIB
fi
cat $O/gemm_product.txt; grep -c . $O/pmc_gemm_summary.txt; cat $O/gemm_probes.txt 2>/dev/null
