#!/bin/bash
# round 5, call E: W3 decode on the streaming kernel (tests + A/B), the reworked in-place cache entries, the pair A/B through bench.py on one more box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
( OMP_NUM_THREADS=24 timeout 400 python -m pytest tests/test_w3.py tests/test_engine_cache.py tests/test_gpu_decode.py -m gpu -q -n 4 -rf --tb=short 2>&1 | grep -v amdgpu.ids | tail -60 ) > $O/pytest.log
grep -E "^FAILED|passed|failed|Error" $O/pytest.log | cut -c1-300 | tail -20
( AWQ_TUNING=1 timeout 240 python tools/w3_decode_ab.py 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/w3_decode_ab.log; cat $O/w3_decode_ab.log
for k in "gemm_v6_pair=0" "gemm_v6_pair=1" "gemm_v6_pair=0" "gemm_v6_pair=1"; do echo -n "$k: "; AWQ_TUNING=1 timeout 120 python bench.py --steps 5 --warmup 2 --no-dropin --no-cpu-baseline --no-batched-decode --no-extra-configs --prefill-m3 0 --prefill-small 0 --tune $k 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['prefill']; print(p['ms_per_pass'], p['roofline']['frac'], 'm4096', d['prefill_m4096']['roofline']['frac'])"; done 2>&1 | tee $O/pair_ab_bench.log
