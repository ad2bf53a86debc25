"""-m gpu: the mid-M kernel (csrc/awq_midm_cdna4.hip; 9 .. 255 rows: the reference's 16 / 32 / 64-row tiles + split_k_iters, gemm_cuda.cu:1155-1206,
:546-619) against the CPU oracle.

* every block shape the library compiles (waves x slabs per wave) x forced K part counts (even and uneven splits, parts of ONE k-step excluded by the
  plan) on a matrix small enough for the oracle to finish in seconds -- plain, with bias, both dtypes, both scale side buffers;
* the Llama-3-8B layer shapes at M = 9, 16, 33, 64, 71, 72, 128, 255 (and 256: the tile kernels' first row count) in bf16 and fp16 through the product
  entry (`awq_w4a16_forward_cdna4` / `_szh`, whatever plan the library picks): the oracle side is computed ONCE per shape for 256 rows of x -- output
  rows are independent, every row count runs on the first m rows;
* the K split is deterministic, needs no initialised workspace, replays from a graph, and two graphs replayed on two streams do not share ticket words."""
import pytest
import torch

from tests.helpers import assert_bits, check_forward_rows, forward_oracle, make_case

pytestmark = pytest.mark.gpu

CFGS = [(8, 1), (4, 1), (4, 2)]  # (waves, slabs per wave)


@pytest.fixture(scope="module")
def env():
    from llm_awq_amd import ops
    ops._capi.lib()
    return ops


def _native(ops, d, K):
    dev = "cuda"
    s, z = d["scales"].to(dev), d["scaled_zeros"].to(dev)
    c4 = ops.repack_v2_to_cdna4(d["qweight"].to(dev))
    szp = ops.pack_sz_cdna4(s, z, K)
    szh, exact = ops.pack_szh_cdna4(s, z, K)
    assert exact
    return c4, s, z, szp, szh


def _reset(ops):
    ops._capi.tune(midm=1, midm_waves=0, midm_ns=0, midm_ks=0, midm_min=65, midm_max=128)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(1296, 1536), (528, 1024)])
def test_every_block_shape_and_part_count(env, dtype, N, K):
    ops = env
    Ms = (9, 17, 33, 64, 65, 100, 128, 200)
    d = make_case(N, K, dtype, seed=N + K, bias=True, M=max(Ms))
    c4, s, z, szp, szh = _native(ops, d, K)
    bias = d["bias"].cuda()
    pre = {False: forward_oracle(d["x"], d["q"], d["scales"], d["scaled_zeros"], dtype),
           True: forward_oracle(d["x"], d["q"], d["scales"], d["scaled_zeros"], dtype, bias=d["bias"])}
    xd = d["x"].cuda()
    try:
        ops._capi.tune(midm_min=9, midm_max=255)  # (the product hands 65 .. 128 rows -- and some shapes' shorter prompts -- to this kernel; here every row count it can serve)
        for (wv, ns) in CFGS:
            for ks in (1, 2, 5):
                ops._capi.tune(midm_waves=wv, midm_ns=ns, midm_ks=ks)
                for M in Ms:
                    for with_bias in (False, True):
                        for side in (None, szh):
                            if side is not None and (with_bias or M % 16 == 0):
                                continue  # (the sz_half form on the uneven row counts, without bias: enough to cover its dequant)
                            y = ops.gemm_cdna4(xd[:M].contiguous(), c4, s, z, bias if with_bias else None, szp, sz_half=side)
                            try:
                                check_forward_rows(y.cpu(), pre[with_bias], M)
                            except AssertionError as e:
                                raise AssertionError(f"waves {wv} ns {ns} ks {ks} M {M} bias {with_bias} szh {side is not None}: {e}") from e
    finally:
        _reset(ops)


LLAMA = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate", 4096, 14336), ("down", 14336, 4096)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name,K,N", LLAMA, ids=[c[0] for c in LLAMA])
def test_llama3_8b_shapes_against_the_oracle(env, dtype, name, K, N):
    ops = env
    if dtype == torch.float16 and name in ("o", "gate"):
        pytest.skip("fp16 on the two extreme shapes (qkv, down); bf16 on all four")
    Ms = (9, 16, 33, 64, 71, 72, 96, 128, 192, 255, 256)
    d = make_case(N, K, dtype, seed=2 * (K * 7 + N), M=256)
    c4, s, z, szp, szh = _native(ops, d, K)
    pre = forward_oracle(d["x"], d["q"], d["scales"], d["scaled_zeros"], dtype)
    xd = d["x"].cuda()
    try:
        for wide in (False, True):  # the product's routing, then every row count on this kernel
            if wide:
                ops._capi.tune(midm_min=9, midm_max=255)
            for M in Ms:
                if wide and 65 <= M <= 128:
                    continue  # (same launch as the first round)
                for side in (None, szh):
                    y = ops.gemm_cdna4(xd[:M].contiguous(), c4, s, z, None, szp, sz_half=side)
                    try:
                        check_forward_rows(y.cpu(), pre, M)
                    except AssertionError as e:
                        raise AssertionError(f"{name} M {M} szh {side is not None} wide {wide}: {e}") from e
    finally:
        _reset(ops)


def test_split_is_deterministic_workspace_free_of_contract_and_graph_safe(env):
    ops = env
    L = ops._capi.lib()
    K, N, M = 4096, 4096, 96
    dtype = torch.bfloat16
    from llm_awq_amd import synth
    w = synth.random_wq(K, N, dtype=dtype, seed=11, keep_q=False)
    c4 = ops.repack_v2_to_cdna4(w["qweight"])
    szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    x = torch.randn(M, K, device="cuda").to(dtype)
    assert L.awq_midm_init() == 0
    wsb = L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K)
    assert wsb > 0 and wsb % (M * N * 4) == 0, "o_proj at 96 rows splits K"

    def run(ws, out, stream=None):
        ops._capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c4.data_ptr(), w["scales"].data_ptr(), w["scaled_zeros"].data_ptr(), szp.data_ptr(), None,
                                                  out.data_ptr(), M, N, K, 128, 1, ws.data_ptr() if ws is not None else None, wsb if ws is not None else 0,
                                                  stream if stream is not None else torch.cuda.current_stream().cuda_stream))

    outs = []
    for fill in (0.0, float("nan"), 1e30):  # the fp32 parts may hold anything
        ws = torch.full((wsb // 4,), fill, dtype=torch.float32, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=dtype)
        run(ws, out)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # without a workspace: the unsplit launch, another association of the same fp32 products
    out_u = torch.empty(M, N, device="cuda", dtype=dtype)
    run(None, out_u)
    torch.cuda.synchronize()
    assert_bits(out_u, outs[0], 0.03, what="split vs unsplit")
    assert ((out_u.float() - outs[0].float()).norm() / outs[0].float().norm()).item() < 1e-3
    # two graphs, each with its own workspace and output, captured one after the other on one stream and replayed CONCURRENTLY on two streams, many times:
    # captured launches own their ticket words
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    graphs, gouts = [], []
    for st in (s1, s1):
        ws = torch.empty(wsb // 4, dtype=torch.float32, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=dtype)
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(8):
                    run(ws, out)
        graphs.append((g, ws))
        gouts.append(out)
    torch.cuda.synchronize()
    for it in range(20):
        for o in gouts:
            o.zero_()
        with torch.cuda.stream(s1):
            graphs[0][0].replay()
        with torch.cuda.stream(s2):
            graphs[1][0].replay()
        torch.cuda.synchronize()
        assert torch.equal(gouts[0], outs[0]) and torch.equal(gouts[1], outs[0]), it


def test_workspace_query_and_plan(env):
    ops = env
    L = ops._capi.lib()
    q = L.awq_w4a16_forward_cdna4_workspace_bytes
    assert q(8, 4096, 4096) == 0                      # decode
    for (m, n, k) in ((72, 4096, 4096), (96, 4096, 14336), (128, 6144, 4096), (48, 8192, 28672), (64, 8192, 8192)):
        b = q(m, n, k)
        assert b % (m * n * 4) == 0 and 2 <= b // (m * n * 4) <= 32, (m, n, k, b)
    ops._capi.tune(midm_max=192)                      # (knob: 129 .. 192 rows as two passes that share the scratch -- the product stops at 128 rows)
    b = q(190, 4096, 4096)
    assert b % (95 * 4096 * 4) == 0 and 2 <= b // (95 * 4096 * 4) <= 32, b
    ops._capi.tune(midm_max=128)
    ops._capi.tune(midm_ks=1)
    try:
        assert q(96, 4096, 4096) == 0
    finally:
        _reset(ops)
