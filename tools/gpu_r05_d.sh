#!/bin/bash
# round 5, call D: the re-modelled tolerances (hull with the accumulation slack), the cache tests, the engine's tests -- then the bench line and the
# block-pair A/B through bench.py on this box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
export AWQ_TEST_STATS=$PWD/$O/test_stats.jsonl
( OMP_NUM_THREADS=24 timeout 500 python -m pytest tests/test_engine_cache.py tests/test_gpu_fused_mlp.py tests/test_gpu_decode.py tests/test_gpu_gemm_v6.py tests/test_gpu_fullsize.py tests/test_w3.py tests/test_fused_norm.py tests/test_moe.py tests/test_gpu_tp_partial.py "tests/test_gpu_oracle_fullsize.py::test_full_shapes_against_the_oracle" -m gpu -q -n 4 -rf --tb=short 2>&1 | grep -v amdgpu.ids | tail -150 ) > $O/pytest.log
unset AWQ_TEST_STATS
grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-300 | tail -30
for k in "gemm_v6_pair=0" "gemm_v6_pair=1" "gemm_v6_pair=0" "gemm_v6_pair=1"; do echo -n "$k: "; AWQ_TUNING=1 timeout 120 python bench.py --steps 5 --warmup 2 --no-dropin --no-cpu-baseline --no-batched-decode --no-extra-configs --prefill-m3 0 --prefill-small 0 --tune $k 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; print(p['ms_per_pass'], p['roofline']['frac'], 'm4096', d['prefill_m4096']['roofline']['frac'])"; done 2>&1 | tee $O/pair_ab_bench.log
( timeout 300 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 ) > $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench.json"))
print("decode", d["value"], d["roofline"]["frac"], "prefill", d["prefill"]["roofline"]["frac"], d["prefill"]["ms_per_pass"])
for k in ("prefill_m4096","prefill_m512","prefill_m64","prefill_m128"):
    print(k, d.get(k,{}).get("roofline",{}).get("frac"), d.get(k,{}).get("ms_per_pass"))
for k in ("w3_llama2_7b","tp70b_world1","moe_mixtral"):
    print(k, json.dumps(d.get(k))[:900])
PY
