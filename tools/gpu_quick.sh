#!/bin/bash
# a short sanity pass on a rebuilt library: smoke + the C-ABI load tests + the tests of this round's new paths
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/q; mkdir -p $O
export TMPDIR=/tmp
( timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log; cat $O/smoke.log
( timeout 250 python -m pytest tests/test_capi_load.py tests/test_gpu_decode.py tests/test_gpu_gemm_v6.py tests/test_gpu_splitk.py -q -x -n 4 -k "not 14336" 2>&1 | tail -3 ) > $O/pytest.log; cat $O/pytest.log
