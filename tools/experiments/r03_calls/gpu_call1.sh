set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -q -m gpu --maxfail=40 --durations=25 2>&1 | tail -80 ) > $O/pytest_gpu_seed0.log
tail -5 $O/pytest_gpu_seed0.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log
cat $O/smoke.log
( timeout 400 python bench.py 2>&1 | grep -E "metric|Error|error|Traceback" | tail -3 ) > $O/bench.log
cut -c1-1500 $O/bench.log
