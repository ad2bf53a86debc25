#!/bin/bash
# round 5, call F: W3 streaming decode with the read-back fixed (tests + A/B), the cache tests, and what the x / scale staging DMAs cost a decode launch
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
( OMP_NUM_THREADS=24 timeout 400 python -m pytest tests/test_w3.py tests/test_engine_cache.py -m gpu -q -n 4 -rf --tb=short 2>&1 | grep -v amdgpu.ids | tail -40 ) > $O/pytest.log
grep -E "^FAILED|passed|failed|Error" $O/pytest.log | cut -c1-300 | tail -12
( AWQ_TUNING=1 timeout 200 python tools/w3_decode_ab.py 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/w3_decode_ab.log; cat $O/w3_decode_ab.log
( AWQ_TUNING=1 AWQ_CDNA4_LIB=$PWD/llm_awq_amd/lib/libawq_cdna4_probes.so timeout 240 python tools/gemvd_xprobe.py 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/gemvd_xprobe.log; cat $O/gemvd_xprobe.log
