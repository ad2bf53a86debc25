#!/bin/bash
# A/B of two builds of libawq_cdna4.so on one box: bench.py decode + prefill legs, alternating.  usage: gpu_ab_lib.sh <tag> <other.so>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$1; mkdir -p $O
LIB=llm_awq_amd/lib/libawq_cdna4.so
cp $LIB /tmp/lib_a.so; cp $2 /tmp/lib_b.so
for rep in 1 2 3; do
  for v in a b; do
    cp /tmp/lib_$v.so $LIB
    ( timeout 300 python bench.py --no-cpu-baseline --no-dropin --prefill-m2 0 --prefill-m3 0 2>&1 | tail -1 ) > $O/bench_${v}_$rep.json
    python - "$O/bench_${v}_$rep.json" "$v$rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:4s} decode {d['value']:8.1f} tok/s frac {d['roofline']['frac']:.4f}   prefill M=2048 {d['prefill']['ms_per_pass']:.2f} ms frac {d['prefill']['roofline']['frac']:.4f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-300:])
PY
  done
done
cp /tmp/lib_a.so $LIB
