#!/bin/bash
# round-3 experiment call 3: shape-based skinny routing of batched decode (parity, M sweep with the product routing, bench legs)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fused_mlp.py -q -m gpu -x -n 4 2>&1 | tail -15 ) > $O/pytest_new.log
tail -3 $O/pytest_new.log
( AWQ_TUNING=1 timeout 400 python tools/decode_m_sweep.py quick 2>&1 | grep -v amdgpu.ids ) > $O/decode_m_sweep.txt
cat $O/decode_m_sweep.txt
( timeout 200 python bench.py --no-prefill --no-cpu-baseline --no-dropin 2>&1 | tail -1 ) > $O/bench_decode.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c3/bench_decode.json").read().strip().splitlines()[-1])
print("M=1", d["value"], d["roofline"]["frac"], "| M=4", d["decode_m4"]["tok_s"], d["decode_m4"]["roofline"]["frac"], "| M=7", d["decode_m7"]["tok_s"], d["decode_m7"]["roofline"]["frac"])
PY
