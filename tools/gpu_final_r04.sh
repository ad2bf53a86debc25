#!/bin/bash
# End-of-round call (round 4): smoke, bench with the driver's flags, the tensor-parallel leg on one rank (world size 1 + the shard shapes of an 8-way
# split), rocprof kernel-trace stats of the bench workload, the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) and the MFMA-busy counters of
# the M = 2048 prefill pass.  The driver's pytest command runs in its own call (tools/gpu_full.sh).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f4; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log
tail -1 $O/smoke.log
( timeout 300 python bench.py 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-600 $O/bench.json
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_TP70B_LAYERS=2 timeout 200 python bench.py --steps 5 --warmup 2 --layers 8 2>$O/bench_tp_world1.err | grep '"metric"' | tail -1 ) > $O/bench_tp_world1.json
cut -c1-400 $O/bench_tp_world1.json
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 AWQ_BENCH_FORCE_TP=1 AWQ_BENCH_SHARD_WORLD=8 AWQ_BENCH_TP70B_LAYERS=4 timeout 200 python bench.py --steps 5 --warmup 2 --layers 8 2>$O/bench_tp_shard8.err | grep '"metric"' | tail -1 ) > $O/bench_tp_shard8.json
cut -c1-300 $O/bench_tp_shard8.json
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 1 2>&1 | tail -3 ) > $O/rocprof_bench.log
python tools/rocpd_stats.py $O/prof_bench/bench_results.db $O/bench_kernel_stats.csv > $O/bench_kernel_stats.txt
head -14 $O/bench_kernel_stats.txt
find $O -name "*.db" -delete
( timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_fetch.log
( timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_write.log
python tools/rocpd_pmc.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt
head -8 $O/pmc_traffic.txt
find $O -name "*.db" -delete
( timeout 240 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE -d $O/pmc_mfma -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 2 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_mfma.log
python tools/pmc_summary.py $O/pmc_mfma $O/pmc_mfma_m2048.txt gemm_cdna4
find $O -name "*.db" -delete
if [ "${1:-}" = "probe" ]; then
( timeout 120 python bench.py --overlap-probe 2 --no-prefill --no-dropin --no-cpu-baseline --no-batched-decode 2>/dev/null | tail -1 | cut -c1-400 ) > $O/bench_overlap2.json
( timeout 120 python bench.py --overlap-probe 3 --no-prefill --no-dropin --no-cpu-baseline --no-batched-decode 2>/dev/null | tail -1 | cut -c1-400 ) > $O/bench_overlap3.json
cat $O/bench_overlap2.json $O/bench_overlap3.json
fi
