#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out/pairprof; mkdir -p $O
for p in 0 1; do
  AWQ_TUNING=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/p$p -o b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --prefill-iters 3 --prefill-m2 0 --prefill-m3 0 --tune gemm_v6_pair=$p > $O/log$p.txt 2>&1
  python tools/rocpd_stats.py $O/p$p/b_results.db $O/stats$p.csv > $O/stats$p.txt
  echo "== pair=$p"; grep -i "gemm\|Name" $O/stats$p.txt | head -12
  tail -1 $O/log$p.txt | cut -c1-200
done
find $O -name "*.db" -delete
