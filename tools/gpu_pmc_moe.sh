#!/bin/bash
# Counters of the grouped (MoE) v6 tile launch next to the dense gate/up launch at M = 4096 (both seven rounds of 256 x 256 tiles): why is a grouped round ~130 us
# and a dense one ~109?  Separate --pmc passes, kernel-trace only beside them.  usage (GPU box): tools/gpu_pmc_moe.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/pmc_moe; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/moe_leg.py <<'PY'
import torch
import bench_extra
from llm_awq_amd import load_engine
eng = load_engine(); dev = torch.device("cuda:0"); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    r = bench_extra.moe_mixtral(eng, dev, s, 3)
print(r["block"])
PY
DENSE="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --no-batched-decode --no-graph --no-extra-configs --prefill-iters 1 --prefill-small 0 --prefill-m 4096 --prefill-m2 0 --prefill-m3 0"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  ( timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/moe_$i -o pmc -- env PYTHONPATH=. python /tmp/moe_leg.py 2>&1 | tail -4 ) > $O/moe_$i.log
  ( timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/dense_$i -o pmc -- $DENSE 2>&1 | tail -2 ) > $O/dense_$i.log
done
python tools/pmc_summary.py $O $O/summary.txt v6
grep -E "moe_gemm_cdna4_v6_kernel|grid=458752" $O/summary.txt | cut -c1-220
