"""-m gpu, needs >= 2 devices (self-skips otherwise): the reference's only multi-GPU mode is single-process layer placement
(awq/entry.py:167-186, accelerate device_map): the SAME kernels are launched on several devices from one process.  Kernels that
use more than 64 KiB of dynamic LDS are opted in per (kernel, device) (csrc/awq_kernels.hpp LdsOptIn): run such shapes on cuda:1
FIRST and cuda:0 second, from two threads as well."""
import threading

import pytest
import torch

from tests.helpers import check_forward, make_case

pytestmark = pytest.mark.gpu


def _run_on(dev, results):
    import llm_awq_amd
    from llm_awq_amd import ops
    eng = llm_awq_amd.load_engine()
    with torch.cuda.device(dev):
        # decode: K = 14336 with 16 waves stages > 100 KiB per block; prefill: the 256 x 256 tile kernel needs 128 KiB
        for (N, K, M) in [(64, 14336, 1), (256, 1024, 300), (512, 4096, 40)]:
            c = make_case(N, K, torch.bfloat16, seed=N + K + M, M=M)
            qw = ops.repack_v2_to_cdna4(c["qweight"].to(dev))
            s, z = c["scales"].to(dev), c["scaled_zeros"].to(dev)
            szp = ops.pack_sz_cdna4(s, z, K)
            y = eng.forward_cdna4(c["x"].to(dev), qw, s, z, szp, None)
            check_forward(y.cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
            if M <= 8:
                szh, exact = ops.pack_szh_cdna4(s, z, K)
                assert exact
                check_forward(ops.decode_cdna4(c["x"].to(dev), qw, szh).cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
    results.append(dev)


def test_large_lds_kernels_on_second_device_first():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    done = []
    _run_on("cuda:1", done)
    _run_on("cuda:0", done)
    ts = [threading.Thread(target=_run_on, args=(d, done)) for d in ("cuda:1", "cuda:0")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(done) == 4
