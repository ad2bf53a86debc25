"""Run under `rocprofv3 --kernel-trace --stats`: a few GEMV configurations, eager launches over rotating
weight copies, so the per-kernel-name average duration (no launch gaps) can be read from the stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, synth  # noqa: E402

L = _capi.lib()
dtype = torch.bfloat16
for (K, N) in [(4096, 14336), (14336, 4096), (4096, 4096)]:
    R = 24
    copies = [synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False) for i in range(R)]
    x = torch.randn(1, K, device="cuda").to(dtype)
    out = torch.empty(1, N, device="cuda", dtype=dtype)
    for cfg in [dict(gemv_probe=2, gemv_probe_blocks=2048), dict(gemv_probe=1), dict(gemv_probe=0),
                dict(gemv_probe=0, gemv_waves=8, gemv_pf=4), dict(gemv_probe=1, gemv_waves=8, gemv_pf=4),
                dict(gemv_probe=0, gemv_waves=16, gemv_pf=4), dict(gemv_probe=1, gemv_waves=16, gemv_pf=4)]:
        _capi.tune(gemv_waves=0, gemv_pf=0, gemv_xlds=1, gemv_probe=0, gemv_order=0)
        _capi.tune(**cfg)
        for rep in range(3):
            for c in copies:
                _capi.check(L.awq_w4a16_gemv(x.data_ptr(), c["qweight"].data_ptr(), c["scales"].data_ptr(),
                                             c["scaled_zeros"].data_ptr(), out.data_ptr(), 1, N, K, 128, 1,
                                             torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    del copies
    torch.cuda.empty_cache()
print("done")
