"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON (CPU) -- authoring-time only.

Usage (in the authoring container, where /root/reference is mounted):
    python oracle/gen_golden.py

The reference has no golden vectors of its own; these fixtures pin the oracle
(oracle/awq_oracle.py) and the product's host code to the reference's actual behaviour:
  * pack_intweight                     awq/quantize/qmodule.py:26-65
  * calculate_zeros_width              awq/quantize/qmodule.py:11-23
  * pseudo_quantize_tensor             awq/quantize/quantizer.py:61-103   (n_bit 4 and 3)
  * WQLinear.from_linear               awq/quantize/qmodule.py:139-199
  * qweight_unpack / packing_v2_from_unpacked / multiply_scale_qzero_negative
                                       tinychat/offline-weight-repacker.py:8-73
  * pseudo-quant Linear forward        awq/quantize/quantizer.py:106-122 (F.linear on CPU, fp32)
The GPU box has no /root/reference: tests only read the .npz files.
bf16 tensors are stored as their int16 bit patterns (numpy has no bf16).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("AWQ_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.contiguous().view(torch.int16).numpy()
    return t.contiguous().numpy()


def main():
    sys.modules.setdefault("awq_inference_engine", types.ModuleType("awq_inference_engine"))
    sys.path.insert(0, REF)
    from awq.quantize.qmodule import WQLinear, pack_intweight, calculate_zeros_width
    from awq.quantize.quantizer import pseudo_quantize_tensor

    spec = importlib.util.spec_from_file_location("ref_repacker", os.path.join(REF, "tinychat", "offline-weight-repacker.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20240607)

    # ---- pack_intweight ------------------------------------------------------------------
    pack = {}
    for i, (N, K) in enumerate([(4, 64), (8, 128), (16, 192), (64, 256), (32, 768)]):
        Q = rng.integers(0, 16, size=(N, K)).astype(np.int32)
        pack[f"q{i}"] = Q.astype(np.uint8)
        pack[f"p{i}"] = pack_intweight(torch.from_numpy(Q), interleave=4, kstride=64).numpy()
    # a structured case: Q[n,k] = (n*5 + k) % 16 makes any index slip visible
    N, K = 16, 128
    Q = ((np.arange(N)[:, None] * 5 + np.arange(K)[None, :]) % 16).astype(np.int32)
    pack["q_struct"] = Q.astype(np.uint8)
    pack["p_struct"] = pack_intweight(torch.from_numpy(Q), 4, 64).numpy()
    np.savez_compressed(os.path.join(OUT, "pack_v2.npz"), **pack)

    # ---- calculate_zeros_width -----------------------------------------------------------
    zw = [(K, G, calculate_zeros_width(K, G)) for K in (128, 768, 1024, 3072, 4096, 8192, 11008, 14336, 28672)
          for G in (128, 64, 32) if K % G == 0]
    np.savez_compressed(os.path.join(OUT, "zeros_width.npz"), table=np.array(zw, dtype=np.int64))

    # ---- pseudo_quantize_tensor + from_linear + fake forward ------------------------------
    torch.manual_seed(1234)
    fl = {}
    cases = [("f16_a", torch.float16, 64, 256, True), ("bf16_a", torch.bfloat16, 64, 256, False),
             ("f16_b", torch.float16, 32, 768, False), ("bf16_b", torch.bfloat16, 48, 1280, True)]
    for name, dt, N, K, has_bias in cases:
        lin = torch.nn.Linear(K, N, bias=has_bias)
        lin.weight.data.normal_(0, 0.02)
        lin = lin.to(dt)
        w0 = lin.weight.data.clone()
        wf, s, z = pseudo_quantize_tensor(lin.weight.data, n_bit=4, zero_point=True, q_group_size=128, get_scale_zp=True)
        lin.weight.data = wf
        q = WQLinear.from_linear(lin, 4, 128, False, s, z)
        x = torch.randn(5, K).to(dt)
        y_fake = torch.nn.functional.linear(x.float(), wf.float(), None if lin.bias is None else lin.bias.float())
        fl[name + "_w0"] = bits(w0)
        fl[name + "_wfake"] = bits(wf)
        fl[name + "_s"] = bits(s)
        fl[name + "_z"] = bits(z)
        fl[name + "_qweight"] = q.qweight.numpy()
        fl[name + "_scales"] = bits(q.scales)
        fl[name + "_scaled_zeros"] = bits(q.scaled_zeros)
        if has_bias:
            fl[name + "_bias"] = bits(q.bias.detach())
        fl[name + "_x"] = bits(x)
        fl[name + "_yfake32"] = y_fake.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "from_linear.npz"), **fl)

    # ---- INT3 grid (pseudo quantisation only; the reference has no packed W3) --------------
    w3 = {}
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
        w = (torch.randn(16, 256) * 0.02).to(dt)
        wf, s, z = pseudo_quantize_tensor(w.clone(), n_bit=3, zero_point=True, q_group_size=128, get_scale_zp=True)
        w3[name + "_w0"], w3[name + "_wfake"], w3[name + "_s"], w3[name + "_z"] = bits(w), bits(wf), bits(s), bits(z)
    np.savez_compressed(os.path.join(OUT, "pseudo_w3.npz"), **w3)

    # ---- v1 -> v2 repacker ----------------------------------------------------------------
    rpk = {}
    for i, (N, K, dt) in enumerate([(16, 256, torch.float16), (8, 1280, torch.bfloat16)]):
        G = K // 128
        Gpad = calculate_zeros_width(K, 128) * 8
        qw1 = torch.from_numpy(rng.integers(-2**31, 2**31, size=(N, K // 8), dtype=np.int64).astype(np.int32))
        qz1 = torch.from_numpy(rng.integers(-2**31, 2**31, size=(N, Gpad // 8), dtype=np.int64).astype(np.int32))
        sc1 = (torch.rand(N, Gpad) * 0.01 + 0.001).to(dt)
        sc1[:, G:] = 0
        unpacked = rp.qweight_unpack(qw1)
        qw2 = rp.packing_v2_from_unpacked(unpacked, 4, 64)
        sz2 = rp.multiply_scale_qzero_negative(sc1, qz1, zp_shift=0).transpose(1, 0).contiguous()
        rpk[f"qw1_{i}"], rpk[f"qz1_{i}"], rpk[f"sc1_{i}"] = qw1.numpy(), qz1.numpy(), bits(sc1)
        rpk[f"unpacked_{i}"] = unpacked.numpy().astype(np.uint8)
        rpk[f"qw2_{i}"], rpk[f"sc2_{i}"], rpk[f"sz2_{i}"] = qw2.numpy(), bits(sc1.transpose(1, 0).contiguous()), bits(sz2)
        rpk[f"dtype_{i}"] = np.array([0 if dt == torch.float16 else 1])
    np.savez_compressed(os.path.join(OUT, "repack_v1_v2.npz"), **rpk)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
