#!/bin/bash
# round 5: the PMC traffic passes of the final build (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only beside them) + the block-pair A/B on this box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f5; mkdir -p $O
export TMPDIR=/tmp
( timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-small 0 --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_fetch.log
( timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-small 0 --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_write.log
python tools/rocpd_pmc.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt
head -12 $O/pmc_traffic.txt
find $O -name "*.db" -delete
bash tools/pair_ab_bench.sh
