#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for k in "gemm_v6_pair=0" "gemm_v6_pair=1"; do echo -n "$k: "; AWQ_TUNING=1 timeout 120 python bench.py --no-dropin --no-cpu-baseline --no-batched-decode --tune $k 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; print(p['ms_per_pass'], p['roofline']['frac'], d['prefill_m4096']['roofline']['frac'], d['prefill_m512']['roofline']['frac'], 'decode', d['value'])"; done; done
