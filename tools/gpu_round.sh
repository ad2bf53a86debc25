#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprof kernel-trace stats.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 ) > $O/smoke.log
( timeout 600 python bench.py 2>&1 | tail -5 ) > $O/bench.log
( timeout 600 python tools/gemv_sweep.py --defaults-only --m 1 4 7 2>&1 | tail -60 ) > $O/gemv_sweep.log
( timeout 600 python tools/gemm_sweep.py --m 64 256 2048 4096 2>&1 | tail -80 ) > $O/gemm_sweep.log
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph 2>&1 | tail -5 ) > $O/rocprof_bench.log
ls -R $O/prof_bench | head -30
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -3; cat $O/bench.log
