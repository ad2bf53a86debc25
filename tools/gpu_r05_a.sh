#!/bin/bash
# round 5, call A: the advisor fixes + re-modelled tolerances on the GPU (with the observed norm-wise values recorded), the bench line with the new legs
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
export AWQ_TEST_STATS=$PWD/$O/test_stats.jsonl
( timeout 420 python -m pytest tests/test_engine_cache.py tests/test_gpu_fused_mlp.py tests/test_gpu_decode.py tests/test_gpu_gemm_v6.py tests/test_gpu_fullsize.py tests/test_w3.py tests/test_fused_norm.py tests/test_moe.py tests/test_gpu_tp_partial.py "tests/test_gpu_oracle_fullsize.py::test_full_shapes_against_the_oracle" -m gpu -q -n 6 2>&1 | tail -25 ) > $O/pytest.log
unset AWQ_TEST_STATS
( timeout 300 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 ) > $O/bench.json
tail -5 $O/pytest.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05a/bench.json"))
print("decode", d["value"], d["roofline"]["frac"], "prefill", d["prefill"]["roofline"]["frac"], d["prefill"]["ms_per_pass"])
for k in ("prefill_m4096","prefill_m512","prefill_m64","prefill_m128"):
    print(k, d.get(k,{}).get("roofline",{}).get("frac"), d.get(k,{}).get("ms_per_pass"))
for k in ("w3_llama2_7b","tp70b_world1","moe_mixtral"):
    print(k, json.dumps(d.get(k))[:600])
PY
