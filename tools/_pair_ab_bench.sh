#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for k in "gemm_v6_pair=0" "gemm_v6_pair_nt=0" "gemm_v6_pair_nt=1"; do echo -n "$k: "; AWQ_TUNING=1 timeout 120 python bench.py --steps 5 --warmup 2 --no-dropin --no-cpu-baseline --no-batched-decode --prefill-m2 0 --prefill-m3 0 --tune $k 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; print(p['ms_per_pass'], p['roofline']['frac'])"; done; done
