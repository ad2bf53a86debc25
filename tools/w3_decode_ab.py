"""GPU A/B: W3A16 decode (M = 1) on the Llama-2-7B shapes -- the LDS-DMA streaming kernel on w3c tiles (awq_gemv_dma.hip, BITS = 3; round 5) against the
register-ring kernel it replaces (knob w3_streaming = 0).  us per launch over rotating weight copies (> the Infinity Cache), fraction of 8 TB/s on the
algorithmic bytes (0.375 B per weight + scales + zeros + x + out).  usage: AWQ_TUNING=1 python tools/w3_decode_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402
from tools.w3_moe_sweep import rand_sz  # noqa: E402


def main():
    dt = torch.bfloat16
    tot = {0: 0.0, 1: 0.0}
    bytes_layer = 0
    for (K, N, mult) in [(4096, 12288, 1), (4096, 4096, 1), (4096, 11008, 2), (11008, 4096, 1)]:
        R = max(8, min(40, (700 << 20) // (N * K * 3 // 8)))
        items = []
        for _i in range(R):
            q = torch.randint(0, 8, (N, K), dtype=torch.uint8, device="cuda")
            s, z = rand_sz(K, N)
            items.append(dict(qw=ops.pack_w3(q), s=s, z=z, szp=ops.pack_sz_cdna4(s, z, K)))
            del q
        by = N * K * 3 // 8 + 4 * (K // 128) * N + 2 * K + 2 * N
        bytes_layer += mult * by
        for M in (1, 4):
            x = torch.randn(M, K, device="cuda").to(dt)
            res = []
            for rnd in range(2):
                for knob in (0, 1):
                    _capi.tune(w3_streaming=knob)
                    us = time_graph(lambda c: ops.forward_w3(x, c["qw"], c["s"], c["z"], c["szp"]), items, reps=4)
                    res.append((knob, us))
            best = {k: min(u for kk, u in res if kk == k) for k in (0, 1)}
            if M == 1:
                for k in (0, 1):
                    tot[k] += mult * best[k]
            print(f"K={K:6d} N={N:6d} M={M}: register ring {best[0]:7.2f} us ({by / best[0] / 1e3 / 80:5.1f} %)   streaming {best[1]:7.2f} us ({by / best[1] / 1e3 / 80:5.1f} %)", flush=True)
        del items
        torch.cuda.empty_cache()
    _capi.tune(w3_streaming=1)
    print(f"layer (qkv, o, gate, up, down) M=1: register ring {tot[0]:7.2f} us = {bytes_layer / tot[0] / 1e3 / 80:5.1f} % of 8 TB/s   streaming {tot[1]:7.2f} us = {bytes_layer / tot[1] / 1e3 / 80:5.1f} %")


if __name__ == "__main__":
    main()
