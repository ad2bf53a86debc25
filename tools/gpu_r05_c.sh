#!/bin/bash
# round 5, call C: (1) the engine's per-wave stamps + stream-only / math-only probes, (2) the prefill block-pair A/B per launch, (3) the tests of
# calls A / B with their failure lines (OMP_NUM_THREADS=24, four workers: eight workers with all cores each took 700 s for 13 %)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
( AWQ_TUNING=1 timeout 200 python tools/mlp_engine_try.py quick 2>&1 | grep -v amdgpu.ids | tail -45 ) > $O/engine_try.log; tail -32 $O/engine_try.log
( AWQ_TUNING=1 timeout 300 python tools/v6_pair_ab.py 2048 2>&1 | grep -v amdgpu.ids | tail -20 ) > $O/v6_pair_ab.log; cat $O/v6_pair_ab.log
export AWQ_TEST_STATS=$PWD/$O/test_stats.jsonl
( OMP_NUM_THREADS=24 timeout 600 python -m pytest tests/test_engine_cache.py tests/test_gpu_fused_mlp.py tests/test_gpu_decode.py tests/test_gpu_gemm_v6.py tests/test_gpu_fullsize.py tests/test_w3.py tests/test_fused_norm.py tests/test_moe.py tests/test_gpu_tp_partial.py "tests/test_gpu_oracle_fullsize.py::test_full_shapes_against_the_oracle" -m gpu -q -n 4 -rf --tb=line 2>&1 | grep -v amdgpu.ids | tail -60 ) > $O/pytest.log
tail -40 $O/pytest.log | cut -c1-400
