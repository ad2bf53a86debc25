#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
( AWQ_TUNING=1 timeout 240 python tools/v6_splitk_ab.py 2>&1 | grep -v amdgpu.ids | tail -30 ) > $O/v6_splitk_ab.txt
cat $O/v6_splitk_ab.txt
