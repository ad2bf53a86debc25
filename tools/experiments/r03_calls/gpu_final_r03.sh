#!/bin/bash
# End-of-round call (round 3): the driver's test command on a fresh box, smoke, bench with the driver's flags, the tensor-parallel leg on
# one rank, rocprof kernel-trace stats of the bench workload; then (time permitting) the W3 / MoE sweep and the PMC traffic passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -q -m gpu --maxfail=30 2>&1 | tail -60 ) > $O/pytest_gpu_seed0.log
tail -4 $O/pytest_gpu_seed0.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log
tail -1 $O/smoke.log
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -E "metric|Error|error|Traceback" | tail -3 ) > $O/bench.log
cut -c1-700 $O/bench.log
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_TP70B_LAYERS=2 timeout 200 python bench.py --steps 5 --warmup 2 --layers 8 2>&1 | tail -2 | cut -c1-1500 ) > $O/bench_tp_world1.log
cut -c1-400 $O/bench_tp_world1.log
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 1 2>&1 | tail -3 ) > $O/rocprof_bench.log
python tools/rocpd_stats.py $O/prof_bench/bench_results.db $O/bench_kernel_stats.csv > $O/bench_kernel_stats.txt
head -12 $O/bench_kernel_stats.txt
find $O -name "*.db" -delete
if [ "${1:-}" = "more" ]; then
( AWQ_TUNING=1 timeout 240 python tools/w3_moe_sweep.py 2>&1 | grep -v amdgpu.ids ) > $O/w3_moe_sweep.txt
tail -12 $O/w3_moe_sweep.txt
( timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_fetch.log
( timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_write.log
python tools/rocpd_pmc.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_traffic.json > /dev/null
find $O -name "*.db" -delete
fi
