"""tools/w3_leg.py -- the W3 bench leg alone (bench_extra.w3_llama2_7b), for A/B runs on one box.  --tune key=value (AWQ_TUNING=1) sets knobs."""
import json
import sys

import torch

import bench_extra
import llm_awq_amd

if __name__ == "__main__":
    eng = llm_awq_amd.load_engine()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sets = [[]] + [a.split(",") for a in sys.argv[1:]]
    for kv in sets:
        if kv:
            from llm_awq_amd import _capi
            _capi.tune(**{e.split("=")[0]: int(e.split("=")[1]) for e in kv})
        r = bench_extra.w3_llama2_7b(eng, dev, st, 20, 5, 6)
        d, p = r["decode_m1"], r["prefill_m2048"]
        print(kv, "decode us/layer fused", d["us_per_layer"], "unfused", d["us_per_layer_five_unfused_calls"], "frac", d["roofline"]["frac"],
              "| prefill ms/layer fused", p["ms_per_layer"], "unfused", p["ms_per_layer_five_unfused_calls"], "frac", p["roofline"]["frac"], flush=True)
