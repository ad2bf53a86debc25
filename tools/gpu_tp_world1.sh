#!/bin/bash
# the tensor-parallel bench leg on ONE rank (world size 1: shards = whole matrices, all-reduces skipped): exercises run_tp_bench end to end
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/tp1; mkdir -p $O
export TMPDIR=/tmp
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_TP70B_LAYERS=${1:-4} timeout 300 python bench.py --steps 10 --warmup 3 2>&1 | tail -40 | cut -c1-3000 ) > $O/bench_tp_world1.log
cat $O/bench_tp_world1.log
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_TP70B_LAYERS=${1:-4} AWQ_BENCH_SHARD_WORLD=8 timeout 300 python bench.py --steps 10 --warmup 3 2>&1 | tail -40 | cut -c1-3000 ) > $O/bench_tp_shard8.log
cat $O/bench_tp_shard8.log
