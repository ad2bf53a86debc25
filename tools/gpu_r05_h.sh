#!/bin/bash
# round 5, call H: kernel-trace of the Mixtral leg (grouped tile launch + tail passes)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/moe_once.py <<PY
import torch, llm_awq_amd, bench_extra
eng = llm_awq_amd.load_engine()
dev = torch.device("cuda", 0)
r = bench_extra.moe_mixtral(eng, dev, torch.cuda.Stream(device=dev), 5)
print(r["w1_w3_fused"]["us"], r["w2"]["us"])
PY
( cd $PWD && PYTHONPATH=$PWD timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_moe -o moe -- python /tmp/moe_once.py 2>&1 | tail -3 ) > $O/rocprof_moe.log
python tools/rocpd_stats.py $O/prof_moe/moe_results.db $O/moe_kernel_stats.csv > $O/moe_kernel_stats.txt
head -14 $O/moe_kernel_stats.txt | cut -c1-260
find $O -name "*.db" -delete
