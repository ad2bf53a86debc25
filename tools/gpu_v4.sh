#!/bin/bash
# GPU check of the hand-scheduled prefill K loop (awq_gemm_v4.hip): parity subset, ubench A/B, bench
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cdna4.py -x -q -m gpu -k "gemm or forward" 2>&1 | tail -5 ) > $O/pytest_v4.log
cat $O/pytest_v4.log
( timeout 300 tools/ubench/gemm_ubench 3 103 2>&1 | cut -c1-100 ) > $O/gemm_v4_ab.log
cat $O/gemm_v4_ab.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -E "metric|Error|error|Traceback" | tail -3 ) > $O/bench_v4.log
cut -c1-1500 $O/bench_v4.log
