#!/bin/bash
# the whole GPU suite under another AWQ_TEST_SEED on four xdist workers (about two minutes): usage tools/gpu_seed_xdist.sh <seed>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
S=${1:-2}
O=gpurun_out/seed$S; mkdir -p $O
export TMPDIR=/tmp
( AWQ_TEST_SEED=$S OMP_NUM_THREADS=24 timeout 400 python -m pytest tests -q -m gpu -n 4 --maxfail=40 2>&1 | tail -25 ) > $O/pytest_gpu_seed${S}_xdist4.log
tail -4 $O/pytest_gpu_seed${S}_xdist4.log
