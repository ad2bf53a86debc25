"""not-gpu: libawq_cdna4.so loads and exports every symbol include/awq_cdna4.h declares; argument
validation (no kernel launch needed) returns the documented codes."""
import ctypes
import os
import re

from llm_awq_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "awq_cdna4.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(awq_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 11
    L = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
        assert n in _capi.SIGNATURES, f"{n} missing from the ctypes signature table"
    assert _capi.lib().awq_abi_version() == 1


def test_argument_validation_without_gpu():
    L = _capi.lib()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    ok = (p16, p16, p16, p16, p16)
    assert L.awq_w4a16_gemv(*ok, 1, 64, 256, 64, 0, None) == -2   # group size
    assert L.awq_w4a16_gemv(*ok, 0, 64, 256, 128, 0, None) == -1  # batch
    assert L.awq_w4a16_gemv(*ok, 17, 64, 256, 128, 0, None) == -1
    assert L.awq_w4a16_gemv(*ok, 1, 64, 256, 128, 7, None) == -3  # dtype
    assert L.awq_w4a16_gemv(*ok, 1, 60, 256, 128, 0, None) == -4  # n % 8
    assert L.awq_w4a16_gemv(*ok, 1, 64, 200, 128, 0, None) == -4  # k % 128
    assert L.awq_w4a16_gemv(p16 + 2, p16, p16, p16, p16, 1, 64, 256, 128, 0, None) == -5
    assert L.awq_w4a16_gemv(None, p16, p16, p16, p16, 1, 64, 256, 128, 0, None) == -6
    assert L.awq_w4a16_gemm(*ok, 32, 64, 256, 64, 1, None, 0, None) == -2
    assert L.awq_unpack_v2(p16, p16, 6, 64, None) == -4
    assert b"batch size" in L.awq_status_string(-1) and b"group size" in L.awq_status_string(-2)
    assert L.awq_w4a16_gemm_workspace_bytes(2048, 4096, 4096) >= 0
    # round 5's entries: the 3-bit fused pair and the 3-bit fp32 partial refuse bad arguments before anything is launched
    gu = L.awq_w3a16_mlp_gate_up_forward  # (x, qweight, sz_packed, out, m, n2, k, group, dtype, workspace, bytes, stream)
    assert gu(None, p16, p16, p16, 1, 64, 256, 128, 1, None, 0, None) == -6
    assert gu(p16, p16, p16, p16, 1, 64, 256, 64, 1, None, 0, None) == -2
    assert gu(p16, p16, p16, p16, 1, 64, 256, 128, 9, None, 0, None) == -3
    assert gu(p16, p16, p16, p16, 0, 64, 256, 128, 1, None, 0, None) == -1
    assert gu(p16, p16, p16, p16, 1, 48, 256, 128, 1, None, 0, None) == -4   # n2 % 32: whole slabs of gate AND up rows
    assert gu(p16, p16, p16, p16, 1, 64, 200, 128, 1, None, 0, None) == -4
    assert gu(p16 + 2, p16, p16, p16, 1, 64, 256, 128, 1, None, 0, None) == -5
    assert L.awq_w3a16_mlp_gate_up_forward_workspace_bytes(8, 22016, 4096) == 0
    pw = L.awq_w3a16_partial  # (x, qweight, sz_packed, out_f32, m, n, k, group, dtype, stream)
    assert pw(p16, None, p16, p16, 1, 64, 256, 128, 1, None) == -6
    assert pw(p16, p16, p16, p16, 1, 64, 256, 32, 1, None) == -2
    assert pw(p16, p16, p16, p16, 1, 64, 256, 128, 5, None) == -3
    assert pw(p16, p16, p16, p16, 0, 64, 256, 128, 1, None) == -4 and pw(p16, p16, p16, p16, 1, 24, 256, 128, 1, None) == -4
    assert pw(p16, p16, p16 + 4, p16, 1, 64, 256, 128, 1, None) == -5


def test_engine_module_exports():
    import llm_awq_amd
    eng = llm_awq_amd.load_engine()
    for name in ("gemv_forward_cuda_new", "gemm_forward_cuda_new"):
        assert callable(getattr(eng, name))
    import sys
    llm_awq_amd.install_as_awq_inference_engine()
    assert sys.modules["awq_inference_engine"] is eng


def test_prefill_routing_and_workspace_rule_without_gpu():
    """awq_w4a16_forward_cdna4_workspace_bytes is pure host logic: which kernel family takes m rows (decode <= 8, skinny <= 64 except where its block shapes under-fill the chip, mid-M 65 .. 128, tiles above)
    and how many K parts / ranges an under-filled launch is split into."""
    L = _capi.lib()
    q = L.awq_w4a16_forward_cdna4_workspace_bytes
    tile = 256 * 128 * 4
    # decode / skinny territory, and launches that fill the chip: no workspace
    for (m, n, k) in ((1, 4096, 4096), (16, 4096, 14336), (32, 4096, 4096), (64, 28672, 4096), (32, 8192, 8192), (64, 10240, 8192), (2048, 4096, 4096), (4096, 28672, 4096)):
        assert q(m, n, k) == 0, (m, n, k)
    # the skinny launch's two K parts where its grid leaves half the chip idle (33 .. 64 rows per pass from K = 4096; 17 .. 32 rows from K = 8192; narrower than
    # ~272 slabs): fp32 [2][rows of a pass][n]
    assert q(64, 4096, 14336) == 2 * 64 * 4096 * 4 and q(40, 4096, 8192) == 2 * 40 * 4096 * 4
    assert q(64, 4096, 4096) == 2 * 64 * 4096 * 4 and q(32, 4096, 14336) == 2 * 32 * 4096 * 4
    assert q(64, 6144, 4096) == 0 and q(64, 4096, 4096 + 128) == 0  # (384 slabs: two parts would be 1.5 rounds of blocks; an odd number of k-steps does not split)
    # 65 .. 128 rows: the mid-M kernel -- K parts that fill the chip with blocks of 8 (4) slabs, fp32 [parts][rows of a pass][n]
    assert q(71, 4096, 14336) == 8 * 71 * 4096 * 4 and q(128, 4096, 14336) == 8 * 128 * 4096 * 4  # down_proj: 32 groups of eight slabs x 8 parts of 14 k-steps
    assert q(128, 4096, 4096) == 4 * 128 * 4096 * 4    # o_proj: eight parts of eight-slab groups would be 4 k-steps each -> four waves: 64 groups x 4 parts of 8
    assert q(96, 6144, 4096) == 4 * 96 * 6144 * 4      # qkv: 48 groups x 4 parts
    assert q(100, 28672, 4096) == 0                    # the gate/up pair fills the chip unsplit (224 groups)
    assert q(160, 28672, 4096) == 0                    # (129 .. 255 rows against a wide n: the tile kernels, full rounds: no scratch)
    # ... and the shorter prompts of shapes the skinny kernel's blocks under-fill (round 6, third session): a very long K against a narrow n (Llama-3-70B down_proj),
    # 33 .. 64 rows against n = 8192 (four-slab skinny blocks = half the chip), 49 .. 64 rows of wide pairs that are not one round of seven-slab blocks
    for (m, n, k) in ((16, 8192, 28672), (64, 8192, 28672), (48, 8192, 8192), (64, 8192, 8192)):
        b = q(m, n, k)
        assert b > 0 and b % (m * n * 4) == 0 and 2 <= b // (m * n * 4) <= 32, (m, n, k, b)
    assert q(64, 57344, 8192) == 0 and q(64, 22016, 4096) == 0   # (wide pairs fill the chip unsplit)
    # under-filled launches of the tile kernels: whole partial tiles, 2..16 K ranges, at least 2 quantisation groups per range
    for (m, n, k) in ((147, 4096, 4096), (190, 4096, 14336), (200, 4096, 14336), (255, 4096, 4096), (256, 4096, 4096), (512, 4096, 14336), (1024, 4096, 4096),
                      (256, 1024, 8192), (300, 6144, 4096)):
        b = q(m, n, k)
        tiles = ((m + 255) // 256) * ((n + 127) // 128)
        assert b > 0 and b % (tiles * tile) == 0, (m, n, k, b)
        ks = b // (tiles * tile)
        assert 2 <= ks <= 16 and (k // 128) // ks >= 2, (m, n, k, ks)
    assert q(256, 4096, 4096 + 64) == 0 and q(256, 4100, 4096) == 0  # shapes the cdna4 GEMM does not take
    # a default process cannot reach the knobs (they re-order sums): AWQ_TUNING=1 opts in
    import os
    os.environ.pop("AWQ_TUNING", None)
    assert L.awq_tune_set(b"gemm_splitk", 0) != 0 and q(256, 4096, 4096) > 0
    os.environ["AWQ_TUNING"] = "1"
    try:
        assert L.awq_tune_set(b"gemm_splitk", 0) == 0 and q(256, 4096, 4096) == 0
        assert L.awq_tune_set(b"gemm_splitk", 5) == 0 and q(256, 4096, 14336) == 32 * 5 * tile
        # knob midm = 0: the round-5 routing (the masked 256-row tile from 72 rows x K >= 8 k; knob gemm_small_m = 0: the skinny launch in 64-row passes, two K parts)
        assert L.awq_tune_set(b"gemm_splitk", 1) == 0 and L.awq_tune_set(b"midm", 0) == 0
        b = q(128, 4096, 14336)
        assert b > 0 and b % (32 * tile) == 0
        assert L.awq_tune_set(b"gemm_small_m", 0) == 0 and q(128, 4096, 14336) == 2 * 64 * 4096 * 4 and q(71, 4096, 14336) == 2 * 36 * 4096 * 4
    finally:
        L.awq_tune_set(b"gemm_splitk", 1)
        L.awq_tune_set(b"gemm_small_m", 1)
        L.awq_tune_set(b"midm", 1)


def test_prefill_tile_plan_for_the_llama3_shapes():
    """host-side query of the tile plan (no GPU): every Llama-3-8B launch at M = 2048 / 4096 fills the 256 CUs in whole rounds or says why not"""
    import ctypes
    from llm_awq_amd import _capi
    L = _capi.lib()

    def plan(m, n, bits=4):
        mode, cols = ctypes.c_int(-1), ctypes.c_int(-1)
        blocks = L.awq_w4a16_gemm_cdna4_plan(m, n, bits, ctypes.byref(mode), ctypes.byref(cols))
        return blocks, mode.value, cols.value

    assert plan(2048, 6144) == (256, 3, 0)        # qkv: 8 x 32 blocks of 256 x 192 = one full round (256-wide: 192 tiles, 75 %)
    assert plan(4096, 6144) == (512, 3, 0)        # two full rounds (256-wide: 384 tiles = 1.5)
    assert plan(2048, 4096) == (256, 1, 0)        # o / down: 256 x 128 blocks, one round
    assert plan(2048, 28672) == (8 * 96 + 8 * 32, 2, 96)   # gate/up: three full rounds of 256-wide + one round of 128-wide
    assert plan(4096, 28672) == (16 * 112, 0, 0)  # seven full rounds of 256-wide
    assert plan(2048, 6144, bits=3)[1] != 3       # the 192-wide blocks are W4 only
    assert plan(100, 4096)[1] == 1 and plan(4, 4096)[0] == 0


def test_block_pair_k_split_routing():
    """host-side query (no GPU): the prefill shapes that run as pairs of 256 x 256 blocks, each summing half of K (awq_gemm_v6.hip) -- 256-wide tiles filling
    3/8 .. 1/2 of the chip in whole XCD shares and K >= 8192; their workspace = one fp32 tile + one 64-byte flag line per pair"""
    from llm_awq_amd import _capi
    L = _capi.lib()
    assert L.awq_w4a16_gemm_cdna4_pair_plan(2048, 4096, 14336) == 1 and L.awq_w4a16_gemm_cdna4_pair_plan(1536, 4096, 14336) == 1
    assert L.awq_w4a16_gemm_cdna4_pair_plan(2048, 4096, 11008) == 1                      # Llama-2-7B down_proj
    assert L.awq_w4a16_gemm_cdna4_pair_plan(1024, 8192, 28672) == 1                      # Llama-3-70B down_proj at 1024 rows: 4 x 32 tiles
    assert L.awq_w4a16_gemm_cdna4_pair_plan(2048, 4096, 4096) == 0                       # o_proj: 16 K tiles per half do not pay for the hand-over
    assert L.awq_w4a16_gemm_cdna4_pair_plan(4096, 4096, 14336) == 0 and L.awq_w4a16_gemm_cdna4_pair_plan(1024, 4096, 14336) == 0
    assert L.awq_w4a16_gemm_cdna4_pair_plan(255, 4096 * 32, 14336) == 0
    assert L.awq_w4a16_forward_cdna4_workspace_bytes(2048, 4096, 14336) == 128 * (256 * 256 * 4 + 64)


def test_batched_decode_routing_for_the_llama3_shapes():
    """host-side query (no GPU): which kernel serves 1 .. 8 rows behind awq_w4a16_decode_cdna4 -- the streaming kernel up to 4 rows and on
    narrow projections, the skinny kernel where the per-slab activation staging would crowd LDS (profiles/r03_decode_m_sweep.txt)"""
    from llm_awq_amd import _capi
    L = _capi.lib()

    def plan(m, n, k, epi=0):
        kern = ctypes.c_int(-1)
        passes = L.awq_w4a16_decode_cdna4_plan(m, n, k, epi, ctypes.byref(kern))
        return passes, kern.value

    # round 6, third session: the skinny kernel stages only the x pieces that hold rows and is ahead from ONE row wherever a CU holds >= 1.5 slabs or K is long (16-wave shape)
    for m in (1, 2, 3, 4):
        assert plan(m, 4096, 4096, 0) == (1, int(m >= 2)) and plan(m, 28672, 4096, 2) == (1, 0), m   # o_proj (one slab per CU: the skinny kernel's 16-wave shape from two rows); the wide gate/up pair: streaming kernel
        assert plan(m, 6144, 4096, 0) == (1, 1) and plan(m, 4096, 14336, 0) == (1, int(m >= 2)), m   # qkv (1.5 slabs per CU): skinny kernel from one row; down_proj (K = 14336, one slab per CU): from two
    for m in (5, 6, 7, 8):
        assert plan(m, 28672, 4096, 2) == (1, 1)          # gate/up pair: 7 slabs per CU
        assert plan(m, 4096, 14336, 0) == (1, 1)          # down_proj
        assert plan(m, 4096, 4096, 0) == (1, 1)           # o_proj
        assert plan(m, 6144, 4096, 0) == (1, 1)
    assert plan(1, 4096, 11008) == (1, 0) and plan(2, 4096, 11008) == (1, 1)   # Llama-2-7B down_proj: one slab per CU, K = 86 groups: 16-wave skinny shape from two rows
    assert plan(4, 5120, 4096) == (1, 0)                                        # (between one and 1.5 slabs per CU: not measured, the streaming kernel as before)
    assert plan(8, 28672, 4096, 1)[1] == 0                # the stacked [gate; up] form has no skinny epilogue
    # Llama-3-70B gate/up pair (K = 8192: 16 KiB of x per row and block): the streaming kernel's eight-wave blocks up to two rows, the skinny kernel from three
    assert plan(1, 8192, 28672) == (1, 1) and plan(1, 10240, 8192) == (1, 1) and plan(1, 8192, 8192) == (1, 1) and plan(1, 12288, 4096) == (1, 1)   # two or more slabs per CU (70B down / qkv / o, 7B qkv): skinny kernel from one row
    assert plan(1, 57344, 8192, 2) == (1, 0) and plan(2, 57344, 8192, 2) == (1, 1) and plan(8, 57344, 8192, 2) == (1, 1)                             # 70B gate/up: from two rows
    assert plan(9, 4096, 4096)[0] == 0 and plan(4, 4100, 4096)[0] == 0 and plan(4, 4096, 4000)[0] == 0
    try:  # forced / disabled (tests and sweeps)
        _capi.tune(decode_skinny_from=9)
        assert plan(7, 4096, 14336)[1] == 0 and plan(7, 4096, 14336)[0] >= 2   # the streaming kernel serves 7 rows of K = 14336 in chunks
        _capi.tune(decode_skinny_from=1)
        assert plan(1, 4096, 4096) == (1, 1)
    finally:
        _capi.tune(decode_skinny_from=0)


def test_narrow_tile_kernel_choice():
    """host-side query: the 256 x 128 blocks run on awq_gemm_v6.hip (two slabs per wave) whenever they are not split along K (round 3:
    profiles/r03_v6_128.txt); short prompts that under-fill the chip keep awq_gemm_v4n.hip's split-K, m < 256 and W3 tiles its unsplit kernel"""
    from llm_awq_amd import _capi
    L = _capi.lib()
    q = L.awq_w4a16_gemm_cdna4_narrow_kernel
    assert q(2048, 4096, 4096, 4, 1, 0) == 1 and q(2048, 4096, 14336, 4, 1, 0) == 1      # o / down at M = 2048: 256 blocks, no split
    assert q(2048, 4096, 4096, 4, 0, 0) == 1 and q(4096, 4096, 4096, 4, 1, 2) == 1
    assert q(512, 4096, 4096, 4, 1, 0) >= 2 and q(512, 4096, 14336, 4, 1, 0) >= 2         # 64 tiles for 256 CUs: split-K (v4n)
    assert q(512, 4096, 4096, 4, 0, 0) == 1                                              # ... but without a workspace: unsplit, v6
    assert q(512, 4096, 4096, 4, 1, 2) == 1                                              # the fused SiLU*mul tail is not split either
    assert q(200, 4096, 4096, 4, 0, 0) == 0 and q(2048, 4096, 4096, 3, 0, 0) == 0         # masked single row tile; W3 tiles
    assert q(4, 4096, 4096, 4, 0, 0) == 0
