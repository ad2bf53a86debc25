#!/bin/bash
# quick iteration: selected parity tests, TP leg on one GPU, bench.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cdna4.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_quick.log
( AWQ_BENCH_FORCE_TP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -3 ) > $O/bench_tp1.log
( timeout 600 python bench.py 2>&1 | tail -2 ) > $O/bench.log
tail -4 $O/pytest_quick.log; cat $O/bench_tp1.log | cut -c1-600; cat $O/bench.log
