#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprof kernel-trace stats, PMC passes.  Outputs under gpurun_out/<tag>/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-round}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 ) > $O/smoke.log
( timeout 900 python bench.py 2>&1 | tail -3 ) > $O/bench.log
tail -1 $O/bench.log > $O/bench.json
if [ "${2:-full}" = "full" ]; then
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin --no-graph --prefill-iters 1 2>&1 | tail -3 ) > $O/rocprof_bench.log
( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-graph --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_fetch.log
( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-graph --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_write.log
python tools/rocpd_stats.py $O/prof_bench/bench_results.db $O/bench_kernel_stats.csv > $O/bench_kernel_stats.txt
python tools/rocpd_pmc.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_traffic.json > /dev/null
find $O -name "*.db" -delete
fi
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/bench.log
