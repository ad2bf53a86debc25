#!/bin/bash
# round-6 development pass for the mid-M kernel: its oracle tests, then the sweep (lagged loop against the in-step loop) on the Llama-3-8B shapes
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/md; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_midm.py -q -x -k "every_block or deterministic" 2>&1 | tail -5 ) > $O/pytest.log; cat $O/pytest.log
( MIDM_SZH=1 timeout 500 python tools/midm_sweep.py ${MIDM_MS:-16 32 64} 2>&1 | tail -80 ) > $O/sweep.txt; cat $O/sweep.txt
