#!/bin/bash
# round 5, call B: the persistent MLP engine (check, timing against two launches, per-phase stamps), then the decode leg both ways, then the
# tests call A could not report on (its pytest was cut by the timeout), with their failure lines
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
( timeout 240 python tools/mlp_engine_try.py quick 2>&1 | grep -v amdgpu.ids | tail -40 ) > $O/engine_try.log; cat $O/engine_try.log
for mode in two one two one; do
  ( timeout 120 python bench.py --steps 20 --warmup 5 --no-prefill --no-dropin --no-cpu-baseline --no-extra-configs --no-batched-decode --mlp-decode $mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['roofline']['frac'], d['ms_per_step'])" ) >> $O/decode_ab.log 2>&1
done
cat $O/decode_ab.log
export AWQ_TEST_STATS=$PWD/$O/test_stats.jsonl
( timeout 700 python -m pytest tests/test_engine_cache.py tests/test_gpu_fused_mlp.py tests/test_gpu_decode.py tests/test_gpu_gemm_v6.py tests/test_gpu_fullsize.py tests/test_w3.py tests/test_fused_norm.py tests/test_moe.py tests/test_gpu_tp_partial.py "tests/test_gpu_oracle_fullsize.py::test_full_shapes_against_the_oracle" -m gpu -q -n 8 -rf --tb=line 2>&1 | grep -v amdgpu.ids | tail -60 ) > $O/pytest.log
tail -45 $O/pytest.log
