#!/bin/bash
# round-3 experiment call: skinny-kernel batched decode (parity + M sweep), co-resident blocks per CU for the gate/up launch, xdist suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_decode.py -q -m gpu -x -k "skinny or ring_configurations" 2>&1 | tail -15 ) > $O/pytest_new.log
tail -3 $O/pytest_new.log
( AWQ_TUNING=1 timeout 400 python tools/decode_m_sweep.py quick 2>&1 | grep -v amdgpu.ids ) > $O/decode_m_sweep.txt
cat $O/decode_m_sweep.txt
run() { tag=$1; shift; ( timeout 200 python bench.py --no-prefill --no-cpu-baseline --no-dropin --no-batched-decode "$@" 2>&1 | tail -1 ) > $O/bench_$tag.json; python - "$O/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {d['value']:8.1f} tok/s  frac {d['roofline']['frac']:.4f}  {d['roofline']['avg_launch_us']:.3f} us/launch")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
( run default_a; run want7_a --tune gemvd_want=7; run want5_a --tune gemvd_want=5; run default_b; run want7_b --tune gemvd_want=7; run want5_b --tune gemvd_want=5 ) > $O/want_ab.txt 2>&1
cat $O/want_ab.txt
( AWQ_TEST_SEED=1 OMP_NUM_THREADS=24 timeout 420 python -m pytest tests -q -m gpu -n 4 --maxfail=40 2>&1 | tail -25 ) > $O/pytest_gpu_seed1_xdist4.log
tail -4 $O/pytest_gpu_seed1_xdist4.log
