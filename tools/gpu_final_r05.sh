#!/bin/bash
# End-of-round call (round 5): the driver's pytest command, smoke, bench with the driver's flags, rocprof kernel-trace stats of the bench workload, the PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only beside them).  Artefacts -> gpurun_out/f5 -> profiles/r05_f_*.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f5; mkdir -p $O
export TMPDIR=/tmp
( [ "${SKIP_PYTEST:-0}" = "1" ] || timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/pytest_gpu_seed0.log
tail -3 $O/pytest_gpu_seed0.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log
tail -1 $O/smoke.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-500 $O/bench.json
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-small 0 --prefill-iters 1 2>&1 | tail -3 ) > $O/rocprof_bench.log
python tools/rocpd_stats.py $O/prof_bench/bench_results.db $O/bench_kernel_stats.csv > $O/bench_kernel_stats.txt
head -16 $O/bench_kernel_stats.txt
find $O -name "*.db" -delete
( timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-small 0 --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_fetch.log
( timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-small 0 --prefill-iters 1 --prefill-m2 0 --prefill-m3 0 2>&1 | tail -3 ) > $O/rocprof_pmc_write.log
python tools/rocpd_pmc.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt
head -10 $O/pmc_traffic.txt
find $O -name "*.db" -delete
