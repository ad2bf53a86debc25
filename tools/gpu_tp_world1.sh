#!/bin/bash
# the tensor-parallel bench leg on ONE rank (world size 1, and the shard shapes of an 8-way split): keeps `bench.py --gpus N`'s code path exercised on the 1-GPU box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/tp1; mkdir -p $O
export TMPDIR=/tmp
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_TP70B_LAYERS=2 timeout 200 python bench.py --steps 5 --warmup 2 --layers 8 2>$O/bench_tp_world1.err | grep '"metric"' | tail -1 ) > $O/bench_tp_world1.json
cut -c1-700 $O/bench_tp_world1.json; tail -3 $O/bench_tp_world1.err
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_SHARD_WORLD=8 AWQ_BENCH_TP70B_LAYERS=4 timeout 200 python bench.py --steps 5 --warmup 2 --layers 8 2>$O/bench_tp_shard8.err | grep '"metric"' | tail -1 ) > $O/bench_tp_shard8.json
cut -c1-500 $O/bench_tp_shard8.json; tail -3 $O/bench_tp_shard8.err
