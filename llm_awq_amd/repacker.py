"""Offline checkpoint repacker -- the MI355X rewrite of tinychat/offline-weight-repacker.py.

    python -m llm_awq_amd.repacker --input vicuna-7b-w4-g128-awq.pt --output vicuna-7b-w4-g128-awq-cdna4.pt
                                   [--target cdna4|v2] [--device cuda:0]

The reference tool (offline-weight-repacker.py:111-152) converts a v1 checkpoint
    <p>.qweight int32 [N, K/8]   <p>.scales T [N, Gpad]   <p>.qzeros int32 [N, Gpad/8]
into the v2 contract of awq/quantize/qmodule.py
    <p>.qweight int16 [N/4, K]   <p>.scales T [Gpad, N]   <p>.scaled_zeros T [Gpad, N]
with Python loops over K on the CPU.  This one accepts v1 OR v2 input, does the nibble shuffles with HIP kernels
through the C ABI (awq_repack_v1_to_v2 / awq_repack_v2_to_cdna4, bit-exact against the reference, see
tests/golden/repack_v1_v2.npz) and by default emits the CDNA4-friendly int4 interleave ("cdna4"): same
keys, shapes and dtypes as v2, `qweight` permuted so that one 16-row x 128-k tile is a contiguous 1-KiB wave load
whose nibbles feed the matrix-core dequant, plus a one-byte marker `<p>.qweight_layout` that
`llm_awq_amd.qmodule.WQLinear` reads at load time.  Everything else in the checkpoint is copied.

Same key rules as the reference: a tensor is a packed weight if "qweight" is in its key, a scale if "scales" is,
"qzeros" keys are consumed with their scales; v2 inputs ("scaled_zeros" present) skip the v1 step.
Layers the cdna4 interleave cannot hold (N % 16 != 0 or K % 128 != 0) stay v2.
"""
from __future__ import annotations

import argparse
from collections import OrderedDict
from typing import Callable, Dict, Optional

import torch


class GpuKernels:
    """The product path: HIP kernels behind the C ABI (no CPU fallback)."""

    def __init__(self, device="cuda"):
        from . import ops
        self.ops = ops
        self.device = torch.device(device)

    def v1_to_v2(self, qweight_v1, scales_v1, qzeros_v1):
        d = self.device
        qw, s, z = self.ops.repack_v1_to_v2(qweight_v1.to(d).contiguous(), scales_v1.to(d).contiguous(),
                                            qzeros_v1.to(d).contiguous())
        return qw.cpu(), s.cpu(), z.cpu()

    def v2_to_cdna4(self, qweight_v2):
        return self.ops.repack_v2_to_cdna4(qweight_v2.to(self.device).contiguous()).cpu()


def cdna4_eligible(qweight_v2: torch.Tensor, scales_v2: torch.Tensor) -> bool:
    n, k = qweight_v2.shape[0] * 4, qweight_v2.shape[1]
    return n % 16 == 0 and k % 128 == 0 and scales_v2.dtype in (torch.bfloat16, torch.float16)


def repack_state_dict(sd: Dict[str, torch.Tensor], target: str = "cdna4", device: str = "cuda",
                      kernels=None, log: Optional[Callable[[str], None]] = None) -> "OrderedDict[str, torch.Tensor]":
    """v1 or v2 state dict -> `target` ("v2" or "cdna4") state dict (CPU tensors)."""
    assert target in ("v2", "cdna4")
    kernels = kernels or GpuKernels(device)
    log = log or (lambda s: None)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    is_v1 = any("qzeros" in k for k in sd)
    for key, param in sd.items():
        assert isinstance(param, torch.Tensor)
        if "qweight_layout" in key:
            continue  # re-derived below
        if "qweight" in key:
            s_key = key.replace("qweight", "scales")
            if is_v1:
                log(f"repacking: {key} (+ {s_key}, qzeros)")
                z_key = key.replace("qweight", "qzeros")
                qw, s, sz = kernels.v1_to_v2(param, sd[s_key], sd[z_key])
                out[s_key], out[key.replace("qweight", "scaled_zeros")] = s, sz
            else:
                qw, s = param, sd[s_key]
                if sd.get(key + "_layout") is not None and int(sd[key + "_layout"]) == 1:
                    raise ValueError(f"{key} is already cdna4-interleaved; convert back with WQLinear.to_v2() first")
            if target == "cdna4" and cdna4_eligible(qw, s):
                log(f"interleaving: {key} -> cdna4")
                qw = kernels.v2_to_cdna4(qw)
                out[key + "_layout"] = torch.tensor(1, dtype=torch.uint8)
            out[key] = qw
        elif is_v1 and ("scales" in key or "qzeros" in key):
            continue  # emitted together with their qweight (offline-weight-repacker.py:129-147)
        else:
            if key not in out:
                log(f"copying: {key}")
                out[key] = param
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--input", type=str, default="./vicuna-7b-w4-g128-awq.pt")
    ap.add_argument("--output", type=str, default="./vicuna-7b-w4-g128-awq-cdna4.pt")
    ap.add_argument("--target", choices=["cdna4", "v2"], default="cdna4")
    ap.add_argument("--device", type=str, default="cuda")
    args = ap.parse_args(argv)
    sd = torch.load(args.input, map_location="cpu")
    out = repack_state_dict(sd, args.target, args.device, log=print)
    torch.save(out, args.output)
    print(f"wrote {args.output} ({len(out)} tensors, target={args.target})")


if __name__ == "__main__":
    main()
