#!/bin/bash
# End-of-round call: bench (default flags) + rocprof kernel-trace stats of the same workload.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python bench.py 2>&1 | grep -E "metric|Error|error|Traceback" | tail -3 ) > $O/bench.log
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --prefill-iters 1 2>&1 | tail -3 ) > $O/rocprof_bench.log
python tools/rocpd_stats.py $O/prof_bench/bench_results.db $O/bench_kernel_stats.csv > $O/bench_kernel_stats.txt
find $O -name "*.db" -delete
cut -c1-3000 $O/bench.log; head -30 $O/bench_kernel_stats.txt
