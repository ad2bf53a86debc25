#!/bin/bash
# round 5, call G: the grouped GEMM's tail pass (partial row tiles of < 64 rows on the grouped skinny kernel): tests, then the Mixtral leg with and without it
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
( OMP_NUM_THREADS=24 timeout 400 python -m pytest tests/test_moe.py -m gpu -q -n 4 -rf --tb=short 2>&1 | grep -v amdgpu.ids | tail -30 ) > $O/pytest.log
grep -E "^FAILED|passed|failed|Error" $O/pytest.log | cut -c1-300 | tail -12
for t in 0 64 0 64; do
AWQ_TUNING=1 timeout 200 python - <<PY 2>&1 | grep -v amdgpu.ids | tee -a $O/moe_tail_ab.log
import json, torch, llm_awq_amd, bench_extra
from llm_awq_amd import _capi
eng = llm_awq_amd.load_engine()
_capi.tune(moe_tail=$t)
dev = torch.device("cuda", 0)
r = bench_extra.moe_mixtral(eng, dev, torch.cuda.Stream(device=dev), 7)
print("moe_tail=$t", r["rows_per_expert"], "w1_w3", r["w1_w3_fused"]["us"], r["w1_w3_fused"]["roofline"]["frac"], "w2", r["w2"]["us"], r["w2"]["roofline"]["frac"], "block", r["block"]["roofline"]["frac"])
PY
done
