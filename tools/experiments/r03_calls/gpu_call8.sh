#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c8; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_gemm_v6.py tests/test_gpu_fullsize.py -q -m gpu -x -k "128_wide or tile_widths or fused_mlp_fullsize" 2>&1 | tail -4 ) > $O/pytest.log
tail -2 $O/pytest.log
run() { tag=$1; shift; ( timeout 200 python bench.py --no-cpu-baseline --no-dropin --no-batched-decode --prefill-m2 0 --prefill-m3 0 --steps 10 --warmup 3 "$@" 2>&1 | tail -1 ) > $O/bench_$tag.json; python - "$O/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:14s} prefill M=2048 {d['prefill']['ms_per_pass']:7.3f} ms frac {d['prefill']['roofline']['frac']:.4f} | decode {d['value']:.1f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
( run hoist0_a --tune gemm_v6_hoist=0; run hoist1_a; run hoist0_b --tune gemm_v6_hoist=0; run hoist1_b; run hoist0_c --tune gemm_v6_hoist=0; run hoist1_c ) > $O/hoist_ab.txt 2>&1
cat $O/hoist_ab.txt
