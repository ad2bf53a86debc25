"""Synthetic WQLinear buffers with the real models' layer shapes (no checkpoints exist in the
sandbox).  Everything is generated ON the GPU and packed by the HIP pack kernel, so full-size
Llama shapes take milliseconds.  Used by bench.py, smoke() and the full-size GPU tests."""
from __future__ import annotations

import torch

from . import ops
from .qmodule import calculate_zeros_width

# (K, N) of the quantised linears of one decoder block, in call order
LLAMA3_8B = {"qkv": (4096, 6144), "o": (4096, 4096), "gate": (4096, 14336), "up": (4096, 14336), "down": (14336, 4096)}
LLAMA2_7B = {"qkv": (4096, 12288), "o": (4096, 4096), "gate": (4096, 11008), "up": (4096, 11008), "down": (11008, 4096)}
LLAMA3_70B = {"qkv": (8192, 10240), "o": (8192, 8192), "gate": (8192, 28672), "up": (8192, 28672), "down": (28672, 8192)}
OPT_125M = {"qkv": (768, 2304), "out": (768, 768), "fc1": (768, 3072), "fc2": (3072, 768)}


def random_wq(K: int, N: int, dtype=torch.bfloat16, device="cuda", seed: int = 0, group_size: int = 128,
              weight_std: float = 0.02, keep_q: bool = True):
    """Random 4-bit integers + per-group (scale, zero) with the statistics of a N(0, std^2) weight
    quantised on the reference's grid (quantizer.py:61-103): range ~ 2*2.9*std per 128-group."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    groups = K // group_size
    gpad = calculate_zeros_width(K, group_size) * 8
    q = torch.randint(0, 16, (N, K), dtype=torch.uint8, device=device, generator=g)
    qweight = ops.pack_v2(q)
    rng = (5.2 + 0.8 * torch.rand(groups, N, device=device, generator=g)) * weight_std
    scales = torch.zeros(gpad, N, dtype=dtype, device=device)
    scales[:groups] = (rng / 15.0).to(dtype)
    zeros = torch.randint(5, 11, (groups, N), device=device, generator=g)
    scaled_zeros = torch.zeros(gpad, N, dtype=dtype, device=device)
    scaled_zeros[:groups] = -(scales[:groups] * zeros.to(torch.float32)).to(dtype)
    return dict(q=q if keep_q else None, qweight=qweight, scales=scales, scaled_zeros=scaled_zeros, K=K, N=N)
