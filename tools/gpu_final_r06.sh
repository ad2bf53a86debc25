#!/bin/bash
# End-of-round call (round 6): the driver's pytest command (SKIP_PYTEST=1 skips it), smoke, bench with the driver's flags, the decode step as a hipGraph replay
# under rocprofv3 (tools/gpu_decode_graph_profile.sh), an eager kernel-trace of the prefill legs, the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs,
# kernel-trace only beside them) of the SAME command, and the matrix-pipe counters of the M = 2048 prefill pass.  Artefacts -> gpurun_out/f6 -> profiles/r06_<tag>_*.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f6; mkdir -p $O
export TMPDIR=/tmp
( [ "${SKIP_PYTEST:-0}" = "1" ] || timeout 1100 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | grep -v amdgpu.ids | tail -24 ) > $O/pytest_gpu_seed0.log
tail -3 $O/pytest_gpu_seed0.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log
tail -1 $O/smoke.log
( timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 ) > $O/bench.json
cut -c1-400 $O/bench.json
bash tools/gpu_decode_graph_profile.sh $O/dg | cut -c1-300
PF="--no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-iters 1"
( timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 20 --warmup 3 $PF 2>&1 | tail -3 ) > $O/rocprof_bench.log
python tools/rocpd_stats.py $O/prof_bench/bench_results.db $O/bench_kernel_stats.csv > $O/bench_kernel_stats.txt
head -24 $O/bench_kernel_stats.txt | cut -c1-220
find $O -name "*.db" -delete
PT="$PF --steps 3 --warmup 1 --prefill-small 0 --prefill-m2 0 --prefill-m3 0"
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python bench.py $PT 2>&1 | tail -3 ) > $O/rocprof_pmc_fetch.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python bench.py $PT 2>&1 | tail -3 ) > $O/rocprof_pmc_write.log
python tools/rocpd_pmc.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt
head -10 $O/pmc_traffic.txt | cut -c1-260
find $O -name "*.db" -delete
( timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_mfma -o pmc -- python bench.py $PT 2>&1 | tail -3 ) > $O/rocprof_pmc_mfma.log
python tools/pmc_summary.py $O/pmc_mfma $O/pmc_mfma_m2048.txt gemm_cdna4
head -16 $O/pmc_mfma_m2048.txt | cut -c1-200
find $O -name "*.db" -delete
