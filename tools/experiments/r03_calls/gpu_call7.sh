#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
( timeout 280 python -m pytest tests/test_w3.py tests/test_gpu_oracle_fullsize.py -q -m gpu -x -k "w3 or W3 or llama2" 2>&1 | tail -6 ) > $O/pytest_w3.log
tail -3 $O/pytest_w3.log
( AWQ_TUNING=1 timeout 200 python tools/w3_moe_sweep.py 2>&1 | grep -v amdgpu.ids | head -10 ) > $O/w3_sweep.txt
cat $O/w3_sweep.txt
