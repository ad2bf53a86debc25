"""-m gpu: the lazy v2 -> cdna4 weight cache behind the drop-in entry points (awq_inference_engine.gemv_forward_cuda_new /
gemm_forward_cuda_new): tinychat passes raw reference-layout buffers, so the extension re-packs them once per tensor
identity + version and must never serve a stale entry."""
import pytest
import torch

from tests.helpers import check_forward, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    import llm_awq_amd
    e = llm_awq_amd.load_engine()
    e.cdna4_cache_enable(True)
    e.cdna4_cache_clear()
    yield e
    e.cdna4_cache_enable(True)
    e.cdna4_cache_clear()


def _dev(c):
    return c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda()


def test_cache_builds_once_and_matches_oracle(eng):
    N, K = 768, 1280
    c = make_case(N, K, torch.bfloat16, seed=1, M=300)
    qw, s, z = _dev(c)
    b0 = eng.cdna4_cache_info()
    for M in (1, 7, 8, 40, 300):
        x = c["x"][:M].contiguous()
        y = (eng.gemv_forward_cuda_new(x.cuda(), qw, s, z, M, N, K, 128) if M < 8 else eng.gemm_forward_cuda_new(x.cuda(), qw, s, z)).cpu()
        check_forward(y, x, c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
    info = eng.cdna4_cache_info()
    assert info["builds"] == b0["builds"] + 1 and info["hits"] >= b0["hits"] + 4 and info["entries"] == 1


def test_in_place_update_and_address_reuse_never_serve_stale_weights(eng):
    N, K = 256, 512
    a, b = make_case(N, K, torch.bfloat16, seed=2, M=4), make_case(N, K, torch.bfloat16, seed=3, M=4)
    qw, s, z = _dev(a)
    x = a["x"].cuda()
    check_forward(eng.gemv_forward_cuda_new(x, qw, s, z, 4, N, K, 128).cpu(), a["x"], a["q"], a["scales"], a["scaled_zeros"], torch.bfloat16)
    # in-place overwrite with other weights: version counters change -> re-pack
    qw.copy_(b["qweight"].cuda())
    s.copy_(b["scales"].cuda())
    z.copy_(b["scaled_zeros"].cuda())
    check_forward(eng.gemv_forward_cuda_new(x, qw, s, z, 4, N, K, 128).cpu(), a["x"], b["q"], b["scales"], b["scaled_zeros"], torch.bfloat16)
    # free and re-allocate: the caching allocator hands the same addresses to NEW tensors
    ptr = qw.data_ptr()
    del qw, s, z
    qw2, s2, z2 = _dev(a)
    y = eng.gemv_forward_cuda_new(x, qw2, s2, z2, 4, N, K, 128).cpu()
    check_forward(y, a["x"], a["q"], a["scales"], a["scaled_zeros"], torch.bfloat16)
    assert eng.cdna4_cache_info()["entries"] == 1, (ptr, qw2.data_ptr())


def test_fp16_is_cached_too_and_disabled_and_capture_bypass_the_cache(eng):
    N, K = 256, 512
    c16 = make_case(N, K, torch.float16, seed=4, M=2)
    qw, s, z = _dev(c16)
    y = eng.gemv_forward_cuda_new(c16["x"].cuda(), qw, s, z, 2, N, K, 128).cpu()
    check_forward(y, c16["x"], c16["q"], c16["scales"], c16["scaled_zeros"], torch.float16)
    assert eng.cdna4_cache_info()["entries"] == 1  # fp16 takes the cdna4 kernels too (matrix-core dequant, offset 1024)
    eng.cdna4_cache_clear()
    c = make_case(N, K, torch.bfloat16, seed=5, M=2)
    qw, s, z = _dev(c)
    eng.cdna4_cache_enable(False)
    y = eng.gemv_forward_cuda_new(c["x"].cuda(), qw, s, z, 2, N, K, 128).cpu()
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
    assert eng.cdna4_cache_info()["entries"] == 0
    eng.cdna4_cache_enable(True)
    # a first call INSIDE a hipGraph capture must not allocate / re-pack: it runs the reference-layout kernels
    x = c["x"].cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            yg = eng.gemv_forward_cuda_new(x, qw, s, z, 2, N, K, 128)
        g.replay()
    torch.cuda.synchronize()
    assert eng.cdna4_cache_info()["entries"] == 0
    check_forward(yg.cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)


def test_warm_entry_is_used_inside_a_capture(eng):
    """tinychat warms up, then captures decode into a graph: the capture must run the cdna4 kernels from the entry the warm-up
    built (a lookup allocates nothing), not fall back to the reference-layout kernels."""
    N, K = 256, 512
    c = make_case(N, K, torch.bfloat16, seed=8, M=2)
    qw, s, z = _dev(c)
    x = c["x"].cuda()
    eng.cdna4_cache_clear()
    y0 = eng.gemv_forward_cuda_new(x, qw, s, z, 2, N, K, 128)  # warm-up: builds the entry
    info0 = eng.cdna4_cache_info()
    assert info0["entries"] == 1
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            yg = eng.gemv_forward_cuda_new(x, qw, s, z, 2, N, K, 128)
        g.replay()
    torch.cuda.synchronize()
    info1 = eng.cdna4_cache_info()
    assert info1["entries"] == 1 and info1["hits"] == info0["hits"] + 1 and info1["builds"] == info0["builds"]
    assert torch.equal(yg, y0)


def test_temporary_zeros_rebuild_only_the_scale_buffer(eng):
    """tinychat's QuantLlamaMLP passes the module's scaled_zeros when decoding and a FRESH `scaled_zeros - 8 * scales` tensor on
    every prefill call (fused_mlp.py:69,76): the weights must be re-packed once, only the 3 %-sized scale buffer per temporary,
    and the decode pair must keep hitting its own slot."""
    N, K = 512, 1024
    c = make_case(N, K, torch.bfloat16, seed=11, M=64)
    qw, s, z = _dev(c)
    xd, xp = c["x"][:2].contiguous(), c["x"]
    eng.cdna4_cache_clear()
    b0 = eng.cdna4_cache_info()
    for it in range(4):
        yd = eng.gemv_forward_cuda_new(xd.cuda(), qw, s, z, 2, N, K, 128).cpu()
        check_forward(yd, xd, c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
        ztmp = z - 8 * s  # a new tensor every call
        yp = eng.gemm_forward_cuda_new(xp.cuda(), qw, s, ztmp).cpu()
        check_forward(yp, xp, c["q"], c["scales"], ztmp.cpu(), torch.bfloat16)
    info = eng.cdna4_cache_info()
    assert info["builds"] == b0["builds"] + 1, info          # the weights: once
    assert info["sz_builds"] == b0["sz_builds"] + 1 + 4, info  # the module's pair once, one per temporary
    assert info["entries"] == 1


def test_in_place_conversion_keeps_no_second_copy_and_can_be_undone(eng):
    """AWQ_CDNA4_INPLACE (here switched on at run time): the qweight is converted where it lies -- cache bytes stay 0, the buffer holds
    the oracle's cdna4 interleave, every row count still matches the oracle, an in-place update of the weights is picked up, and
    cdna4_restore / cdna4_cache_clear / switching the cache off put the reference interleave back bit for bit."""
    import numpy as np
    from oracle import awq_oracle as O
    N, K = 512, 1280
    a, b = make_case(N, K, torch.bfloat16, seed=12, M=40), make_case(N, K, torch.bfloat16, seed=13, M=40)
    try:
        eng.cdna4_cache_inplace(True)
        qw, s, z = _dev(a)
        v2 = qw.clone()
        for M in (1, 7, 8, 40):
            x = a["x"][:M].contiguous()
            y = (eng.gemv_forward_cuda_new(x.cuda(), qw, s, z, M, N, K, 128) if M < 8 else eng.gemm_forward_cuda_new(x.cuda(), qw, s, z)).cpu()
            check_forward(y, x, a["q"], a["scales"], a["scaled_zeros"], torch.bfloat16)
        info = eng.cdna4_cache_info()
        assert info["bytes"] == 0 and info["entries"] == 1 and info["inplace"]
        assert np.array_equal(qw.cpu().numpy(), O.pack_cdna4(a["q"])), "the module's buffer now holds the cdna4 interleave"
        # new reference-layout weights copied over it (load_state_dict does this): converted again, never served stale
        qw.copy_(b["qweight"].cuda())
        s.copy_(b["scales"].cuda())
        z.copy_(b["scaled_zeros"].cuda())
        x = a["x"][:4].contiguous()
        check_forward(eng.gemv_forward_cuda_new(x.cuda(), qw, s, z, 4, N, K, 128).cpu(), x, b["q"], b["scales"], b["scaled_zeros"], torch.bfloat16)
        assert eng.cdna4_restore(qw) is True
        assert torch.equal(qw.cpu(), b["qweight"]), "restored to the reference interleave"
        assert eng.cdna4_restore(qw) is False
        # converted again on the next call; clearing the cache restores every live in-place entry
        eng.gemv_forward_cuda_new(x.cuda(), qw, s, z, 4, N, K, 128)
        assert not torch.equal(qw.cpu(), b["qweight"])
        eng.cdna4_cache_clear()
        assert torch.equal(qw.cpu(), b["qweight"])
        eng.gemv_forward_cuda_new(x.cuda(), qw, s, z, 4, N, K, 128)
        eng.cdna4_cache_enable(False)   # the reference-layout kernels will read it: must be v2 again
        assert torch.equal(qw.cpu(), b["qweight"])
        check_forward(eng.gemv_forward_cuda_new(x.cuda(), qw, s, z, 4, N, K, 128).cpu(), x, b["q"], b["scales"], b["scaled_zeros"], torch.bfloat16)
        del v2
    finally:
        eng.cdna4_cache_enable(True)
        eng.cdna4_cache_inplace(False)


def test_in_place_entry_follows_aliases_and_refuses_what_it_cannot_serve(eng):
    """with the in-place mode on, ANOTHER tensor over the converted bytes (`.detach()`, `.data`, a view of the whole buffer) must
    hit the same entry -- not be taken for v2 data and permuted a second time -- `cdna4_is_converted` / `WQLinear.engine_converted()` report
    the state to format tools (which raise), a call the cdna4 kernels cannot serve (fp32 scales) raises instead of running the v2 kernels
    on permuted bytes, and the restore puts the reference interleave back whatever alias it is called through."""
    import numpy as np
    from oracle import awq_oracle as O
    from llm_awq_amd.parallel import TPWQLinear
    from llm_awq_amd.qmodule import WQLinear
    N, K = 256, 1280
    c = make_case(N, K, torch.bfloat16, seed=31, M=8)
    try:
        eng.cdna4_cache_inplace(True)
        qw, s, z = _dev(c)
        x = c["x"][:4].contiguous().cuda()
        assert eng.cdna4_is_converted(qw) is False
        builds0 = eng.cdna4_cache_info()["builds"]
        y0 = eng.gemv_forward_cuda_new(x, qw, s, z, 4, N, K, 128)
        assert eng.cdna4_is_converted(qw) is True
        c4 = O.pack_cdna4(c["q"])
        for alias in (qw.detach(), qw.data, qw.view(N // 4, K), qw[:]):
            y = eng.gemv_forward_cuda_new(x, alias, s, z, 4, N, K, 128)
            assert torch.equal(y, y0)
            assert np.array_equal(qw.cpu().numpy(), c4), "permuted exactly once"
            assert eng.cdna4_is_converted(alias) is True
        check_forward(y0.cpu(), c["x"][:4], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
        assert eng.cdna4_cache_info()["builds"] == builds0 + 1
        # a module over these buffers: the format tools refuse them while they are converted
        lin = WQLinear(4, 128, K, N, False, "cuda", dtype=torch.bfloat16)
        lin.qweight, lin.scales, lin.scaled_zeros = qw, s, z
        assert lin.engine_converted()
        with pytest.raises(RuntimeError, match="cdna4_restore"):
            lin.to_cdna4()
        with pytest.raises(RuntimeError, match="cdna4_restore"):
            TPWQLinear(lin, "row", world=2, rank=0)
        # fp32 scales cannot run on the cdna4 kernels: an error, not the v2 kernels on permuted bytes
        with pytest.raises(RuntimeError, match="cdna4_restore"):
            eng.gemm_forward_cuda_new(x.float().repeat(3, 1), qw, s.float(), z.float())
        # restore through an alias
        assert eng.cdna4_restore(qw.detach()) is True
        assert torch.equal(qw.cpu(), c["qweight"]) and not lin.engine_converted()
        lin.to_cdna4()
        check_forward(lin(x).cpu(), c["x"][:4], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
    finally:
        eng.cdna4_cache_enable(True)
        eng.cdna4_cache_inplace(False)


def test_in_place_entry_overwritten_then_called_through_an_alias(eng):
    """convert in place, overwrite the buffer with fresh v2 bytes (`qw.copy_`, what load_state_dict does), then call through an
    ALIAS (`.detach()`): the alias shares the version counter, the entry is stale -- reconverted, never followed (the cdna4 kernels on v2
    bytes would be silently wrong).  The same state through cdna4_is_converted / cdna4_restore / cdna4_cache_clear: overwritten bytes are v2
    data, nothing is "restored" over them."""
    N, K = 256, 1280
    a, b = make_case(N, K, torch.bfloat16, seed=41, M=4), make_case(N, K, torch.bfloat16, seed=42, M=4)
    try:
        eng.cdna4_cache_inplace(True)
        qw, s, z = _dev(a)
        x = a["x"].cuda()
        check_forward(eng.gemv_forward_cuda_new(x, qw, s, z, 4, N, K, 128).cpu(), a["x"], a["q"], a["scales"], a["scaled_zeros"], torch.bfloat16)
        assert eng.cdna4_is_converted(qw) is True
        # fresh v2 bytes over the converted ones; first consumer is an alias
        qw.copy_(b["qweight"].cuda())
        s.copy_(b["scales"].cuda())
        z.copy_(b["scaled_zeros"].cuda())
        assert eng.cdna4_is_converted(qw.detach()) is False, "overwritten: the buffer holds v2 data again"
        y = eng.gemv_forward_cuda_new(x, qw.detach(), s, z, 4, N, K, 128).cpu()
        check_forward(y, a["x"], b["q"], b["scales"], b["scaled_zeros"], torch.bfloat16)
        assert eng.cdna4_is_converted(qw) is True
        # overwrite again, then restore / clear: the fresh v2 bytes must come through untouched
        qw.copy_(a["qweight"].cuda())
        assert eng.cdna4_restore(qw.detach()) is False
        assert torch.equal(qw.cpu(), a["qweight"])
        s.copy_(a["scales"].cuda())
        z.copy_(a["scaled_zeros"].cuda())
        eng.gemv_forward_cuda_new(x, qw, s, z, 4, N, K, 128)
        qw.copy_(b["qweight"].cuda())
        eng.cdna4_cache_clear()
        assert torch.equal(qw.cpu(), b["qweight"])
    finally:
        eng.cdna4_cache_enable(True)
        eng.cdna4_cache_inplace(False)


def test_in_place_entries_release_their_storage_when_the_model_is_deleted(eng):
    """an in-place entry holds the caller's tensor; once every other holder is gone the next sweep (a cache miss) must drop it so the
    weights go back to the allocator -- the mode exists to save memory."""
    N, K = 2048, 2048
    a, b = make_case(N, K, torch.bfloat16, seed=43, M=2), make_case(256, 512, torch.bfloat16, seed=44, M=2)
    try:
        eng.cdna4_cache_inplace(True)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        qw, s, z = _dev(a)
        x = a["x"].cuda()
        eng.gemv_forward_cuda_new(x, qw, s, z, 2, N, K, 128)
        # a live alias keeps the entry (and its bytes) alive across a sweep
        alias = qw.detach()
        del qw
        qb, sb, zb = _dev(b)
        eng.gemv_forward_cuda_new(b["x"].cuda(), qb, sb, zb, 2, 256, 512, 128)   # miss -> sweep
        assert eng.cdna4_cache_info()["entries"] == 2, ("the entry of a live alias must survive a sweep", eng.cdna4_cache_info())
        y = eng.gemv_forward_cuda_new(x, alias, s, z, 2, N, K, 128).cpu()
        check_forward(y, a["x"], a["q"], a["scales"], a["scaled_zeros"], torch.bfloat16)
        held = torch.cuda.memory_allocated()
        del alias, s, z, x, y
        qc, sc, zc = _dev(make_case(256, 512, torch.bfloat16, seed=45, M=2))
        eng.gemv_forward_cuda_new(b["x"].cuda(), qc, sc, zc, 2, 256, 512, 128)   # miss -> sweep drops the orphaned entry
        torch.cuda.synchronize()
        assert eng.cdna4_cache_info()["entries"] == 2, ("the orphaned entry must be dropped by the sweep", eng.cdna4_cache_info())
        assert torch.cuda.memory_allocated() <= held - N * K // 2 + (1 << 20), ("storage not released", base, held, torch.cuda.memory_allocated())
    finally:
        eng.cdna4_cache_enable(True)
        eng.cdna4_cache_inplace(False)
