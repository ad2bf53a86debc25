#!/bin/bash
# A/B of the block-pair K split through bench.py on whatever box this lands on (profiles/r05_v6_pair.txt): prefill M = 2048 ms per pass and fraction,
# M = 4096 fraction beside it as the box's own reference.  usage: gpurun -- 'bash tools/pair_ab_bench.sh'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pairab
for r in 1 2; do for k in "gemm_v6_pair=0" "gemm_v6_pair=1"; do echo -n "$k: "; AWQ_TUNING=1 timeout 120 python bench.py --steps 5 --warmup 2 --no-dropin --no-cpu-baseline --no-batched-decode --no-extra-configs --prefill-m3 0 --prefill-small 0 --tune $k 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; print(p['ms_per_pass'], p['roofline']['frac'], 'm4096', d['prefill_m4096']['roofline']['frac'], 'decode', d['value'])"; done; done | tee -a gpurun_out/pairab/log_$(date +%s).txt
