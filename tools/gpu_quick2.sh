#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 tools/ubench/gemm_ubench 4 3 2>&1 ) > $O/gemm_ubench.log
( AWQ_BENCH_FORCE_TP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -E "metric|Error|error|Traceback" | tail -3 ) > $O/bench_tp1.log
cat $O/bench_tp1.log | cut -c1-900
