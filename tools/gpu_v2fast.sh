#!/bin/bash
# GPU check of the reference-layout pipelined decode kernel: parity subset, then old-vs-new timing for fp16 and bf16.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_engine_cache.py -x -q -m gpu -k "gemv or forward or golden or cache" 2>&1 | tail -5 ) > $O/pytest_v2fast.log
cat $O/pytest_v2fast.log
( timeout 600 python tools/gemv_sweep.py --defaults-only --dtype f16 --m 1 4 8 2>&1 ) > $O/gemv_v2fast_f16.log
( timeout 600 python tools/gemv_sweep.py --defaults-only --dtype bf16 --m 1 2>&1 ) > $O/gemv_v2fast_bf16.log
cat $O/gemv_v2fast_f16.log $O/gemv_v2fast_bf16.log
