#!/bin/bash
# round-3 experiment call 4: the 128-wide tiles of o_proj / down_proj at M = 2048 on the v6 loop (two slabs per wave) against awq_gemm_v4n.hip
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_gemm_v6.py -q -m gpu -x -k "128_wide" 2>&1 | tail -6 ) > $O/pytest_new.log
tail -3 $O/pytest_new.log
run() { tag=$1; shift; ( timeout 200 python bench.py --no-cpu-baseline --no-dropin --no-batched-decode --prefill-m3 0 --steps 10 --warmup 3 "$@" 2>&1 | tail -1 ) > $O/bench_$tag.json; python - "$O/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:14s} prefill M=2048 {d['prefill']['ms_per_pass']:7.3f} ms frac {d['prefill']['roofline']['frac']:.4f} | M=4096 {d['prefill_m4096']['ms_per_pass']:7.3f} ms frac {d['prefill_m4096']['roofline']['frac']:.4f} | decode {d['value']:.1f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
( run v4n_a; run v6_128_a --tune gemm_v6_128=1; run v4n_b; run v6_128_b --tune gemm_v6_128=1 ) > $O/v6_128_ab.txt 2>&1
cat $O/v6_128_ab.txt
