#!/bin/bash
# PMC counters of the mid-M kernel (and of the round-5 kernels on the same shapes: the sweep's "round 5" leg) -- three passes of 8 SQ counters.
# usage (GPU box): tools/gpu_pmc_midm.sh "<M list>" [tag]     env MIDM_SHAPES / MIDM_CFGS / MIDM_KS / MIDM_SZH pass through
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp MIDM_EAGER=1
MS="${1:-64}"; TAG="${2:-midm}"
run() { ( timeout 600 rocprofv3 --kernel-trace --pmc $2 -d $O/pmc_${TAG}_$1 -o pmc -- python tools/midm_sweep.py $MS 2>&1 | tail -3 ) > $O/pmc_${TAG}_$1.log; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run b "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
run c "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"
run f "FETCH_SIZE"
python tools/pmc_summary.py $O $O/pmc_${TAG}_summary.txt awq::
find $O -name "*.db" -delete
