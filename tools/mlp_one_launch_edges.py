"""Where the time of the one-launch QuantLlamaMLP goes (awq_w4a16_mlp_decode_cdna4): the launch with 1 / 16 / 64 / 256 down_proj blocks behind the 896 gate/up
blocks, against the two launches and the gate/up launch alone -- all replayed from hipGraphs over rotating weight copies (tools/gemvc_sweep.time_graph).
Record: profiles/r04_mlp_one_launch.txt.  usage: python tools/mlp_one_launch_edges.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from llm_awq_amd import ops, _capi
from tools.mlp_one_launch_try import build, one, two
from tools.gemvc_sweep import time_graph
x = torch.randn(1, 4096, device="cuda").bfloat16()
for n_out in (16, 256, 1024, 4096):
    cs = [build(4096, 14336, n_out, 100 + 3 * i) for i in range(8)]
    t1 = time_graph(lambda c: one(c, x), cs)
    t2 = time_graph(lambda c: two(c, x), cs)
    tg = time_graph(lambda c: ops.mlp_gate_up_forward_cdna4(x, c["gu"], c["gu_szp"], c["gu_szh"]), cs)
    print(f"n_out {n_out:5d}: one launch {t1:6.2f} us   two launches {t2:6.2f} us   gate/up launch alone {tg:6.2f} us", flush=True)
    del cs
    torch.cuda.empty_cache()
