#!/bin/bash
# round-3 experiment call 5: column-panel tile walk of the v6 prefill kernel (knob gemm_v6_walk) -- equality + prefill A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
( AWQ_TUNING=1 timeout 120 python - <<'PY'
import torch
from llm_awq_amd import ops, synth, _capi
for (K, N, M) in ((4096, 6144, 2048), (1024, 1296, 777), (4096, 4096, 2048)):
    w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=K + N, keep_q=False)
    c4 = ops.repack_v2_to_cdna4(w["qweight"]); szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    x = torch.randn(M, K, device="cuda").bfloat16()
    y0 = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
    _capi.tune(gemm_v6_walk=1)
    y1 = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
    _capi.tune(gemm_v6_walk=0)
    print("walk equal", (K, N, M), torch.equal(y0, y1))
PY
) 2>&1 | grep -v amdgpu.ids > $O/walk_equal.txt
cat $O/walk_equal.txt
run() { tag=$1; shift; ( timeout 200 python bench.py --no-cpu-baseline --no-dropin --no-batched-decode --prefill-m3 0 --steps 10 --warmup 3 "$@" 2>&1 | tail -1 ) > $O/bench_$tag.json; python - "$O/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:14s} prefill M=2048 {d['prefill']['ms_per_pass']:7.3f} ms frac {d['prefill']['roofline']['frac']:.4f} | M=4096 {d['prefill_m4096']['ms_per_pass']:7.3f} ms frac {d['prefill_m4096']['roofline']['frac']:.4f} | decode {d['value']:.1f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
( run walk0_a; run walk1_a --tune gemm_v6_walk=1; run walk0_b; run walk1_b --tune gemm_v6_walk=1 ) > $O/walk_ab.txt 2>&1
cat $O/walk_ab.txt
