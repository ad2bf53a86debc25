"""RMSNorm in front of a WQLinear -- the MI355X build of the step tinychat takes as two launches: FTLlamaRMSNorm
(tinychat/modules/fused_norm.py:7-21 -> awq/kernels/csrc/layernorm/layernorm.cu:39-61, `layernorm_forward_cuda`) followed by
WQLinear.forward.

`RMSNormWQLinear(norm_weight, eps, linear)` computes  linear(T((float(x) * rsqrt(mean(x^2) + eps)) * float(gamma)))  -- the
kernel's arithmetic: fp32 sum of squares, rsqrtf(variance / n + eps), the two multiplies in fp32 in that order, ONE rounding
to T -- as one launch for decode rows (<= 4): every wave of the decode GEMV sums the squares of the k-slices it stages anyway,
so the 8 KiB activation row never makes a round trip through HBM / L2.  Measured (profiles/r01_norm_fusion.txt): it costs what
the separate norm launch costs for narrow projections and less for the gate/up pair, i.e. it removes a launch from the chain
without removing time from it; it is offered for callers that want the launch count down, it is not the default of anything.
More rows run the norm as one launch of the same arithmetic (`awq_rmsnorm`, csrc/awq_util.hip), then the linear.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import load_engine


class RMSNormWQLinear(nn.Module):
    def __init__(self, norm_weight: torch.Tensor, eps: float, linear):
        super().__init__()
        self.weight = norm_weight            # FTLlamaRMSNorm.weight
        self.variance_epsilon = eps
        self.linear = linear                 # a WQLinear (cdna4 layout for the fused path)

    @torch.no_grad()
    def forward(self, x):
        lin = self.linear
        rows = x.numel() // x.shape[-1]
        if rows <= 4 and getattr(lin, "layout", "v2") == "cdna4" and x.is_cuda and lin.in_features <= 16384:
            if lin.sz_cdna4 is None:
                lin.sz_cdna4 = load_engine().pack_sz_cdna4(lin.scales, lin.scaled_zeros, lin.in_features)
            return load_engine().rmsnorm_forward_cdna4(x.contiguous(), self.weight, float(self.variance_epsilon), lin.qweight,
                                                       lin.sz_cdna4, lin.bias, False)
        # more rows: the norm as ONE launch of the same arithmetic (csrc/awq_util.hip rmsnorm_kernel), then the linear.  No torch
        # arithmetic stands in for the kernel: what it does not take (CPU tensors, fp32 activations, a norm weight of another dtype,
        # k % 8 != 0) raises, like every other entry of this package
        if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and self.weight.dtype == x.dtype and x.shape[-1] % 8 == 0):
            raise RuntimeError("RMSNormWQLinear runs the HIP kernels only: x must be a GPU fp16 / bf16 tensor, the norm weight of the same "
                               "dtype, and the hidden size a multiple of 8 (there is no CPU / PyTorch fallback)")
        return lin(load_engine().rmsnorm(x.contiguous(), self.weight.contiguous(), float(self.variance_epsilon)))
