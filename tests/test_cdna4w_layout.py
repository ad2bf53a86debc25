"""CPU: the PLANNED 32-row interleave "cdna4w" (oracle/awq_oracle.py, DESIGN.md "Next steps after round 3") -- no kernel reads it yet.

What is pinned here so that the kernel work starts from a verified layout:
  * the index map is a bijection of the nibbles (pack / unpack round trip, a pure permutation of the cdna4 buffer's nibbles);
  * a REGISTER-LEVEL emulation of the matrix-core dequant -- the same instruction sequence the shipped kernels run on cdna4 tiles
    (extractions (w >> 4 i) & 0x000F000F, v_mfma_f32_4x4x4_16B with the per-block diagonal B operand, pack of the two results) --
    lands, on cdna4w tiles, with lane (n = l % 32, kb = l / 32) holding k = 16 a + 8 kb + 0..7 of row n: the A operand of
    v_mfma_f32_32x32x16.  The emulator is first held against the SHIPPED cdna4 layout (lane (n = l % 16, g = l / 16), k = 32 a + 8 g +
    0..7: the 16x16x32 operand, which the GPU tests verify on hardware), so its model of the 4x4x4 blocks is the one the hardware has.
"""
import numpy as np
import pytest

from oracle import awq_oracle as O


def _blocks_dequant(words, s_lane, c_lane):
    """words u32 [64 lanes]; s_lane / c_lane float64 [64] (this lane's scale and offset).  Emulates, per 4-lane block of
    v_mfma_f32_4x4x4_16B: A[m][kk] = lane m's four extracted values, B[kk][n] = s_n [kk == n] (the diagonal operand the kernels build
    with v_perm / masks by lane % 4), C = c_n; D[m][n] lands in lane n, register m.  Returns [64 lanes][8] = pack8(d0, d1)."""
    out = np.zeros((64, 8))
    for half, shifts in enumerate(((0, 4), (8, 12))):          # a0 = extractions i = 0, 1; a1 = i = 2, 3
        A = np.zeros((64, 4))
        for e, sh in enumerate(shifts):
            v = (words >> np.uint32(sh)) & np.uint32(0x000F000F)
            A[:, 2 * e] = (v & 0xFFFF).astype(np.float64)       # low half of the register: nibble i
            A[:, 2 * e + 1] = (v >> 16).astype(np.float64)      # high half: nibble i + 4
        for b in range(16):
            for n in range(4):                                  # output lane 4 b + n
                for m in range(4):                              # its register m <- lane 4 b + m, element kk = n
                    out[4 * b + n, 4 * half + m] = A[4 * b + m, n] * s_lane[4 * b + n] + c_lane[4 * b + n]
    return out


@pytest.mark.parametrize("N,K", [(32, 128), (64, 256), (96, 384)])
def test_round_trip_and_permutation_of_the_cdna4_nibbles(N, K):
    rng = np.random.default_rng(N + K)
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    qw = O.pack_cdna4w(q)
    assert qw.shape == (N // 4, K) and qw.dtype == np.int16
    assert np.array_equal(O.unpack_cdna4w(qw), q)
    # every (word, nibble) slot is hit exactly once
    nn, kk = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
    word, p = O.cdna4w_position(nn, kk, K)
    slots = word.astype(np.int64) * 8 + p
    assert np.unique(slots).size == N * K and slots.min() == 0 and slots.max() == N * K - 1
    # a slab PAIR's K extent is one contiguous run of 1-KiB tiles (what a streaming decode kernel needs)
    first = word[0:32].min() // 256, word[0:32].max() // 256
    assert first == (0, K // 64 - 1)


@pytest.mark.parametrize("layout", ["cdna4", "cdna4w"])
def test_matrix_core_dequant_emits_the_product_operand(layout):
    """register-level emulation on both layouts: cdna4 -> the 16x16x32 operand (what the hardware tests confirm), cdna4w -> 32x32x16"""
    N, K = 64, 256
    rng = np.random.default_rng(7)
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    s = rng.uniform(0.004, 0.01, size=(N, K // 128))
    z = -s * rng.integers(0, 16, size=(N, K // 128))
    W = q * np.repeat(s, 128, axis=1) + np.repeat(z, 128, axis=1)          # [N, K] exact in float64
    if layout == "cdna4":
        buf = np.ascontiguousarray(O.pack_cdna4(q)).view(np.uint32).reshape(N // 16, K // 128, 64, 4)
        rows, kspan = 16, 32      # operand: lane (n = l % 16, g = l / 16), word a covers k = 32 a + 8 g + 0..7
    else:
        buf = np.ascontiguousarray(O.pack_cdna4w(q)).view(np.uint32).reshape(N // 32, K // 64, 64, 4)
        rows, kspan = 32, 16      # operand: lane (n = l % 32, kb = l / 32), word a covers k = 16 a + 8 kb + 0..7
    lanes = np.arange(64)
    for nb in range(buf.shape[0]):
        for kt in range(buf.shape[1]):
            k0 = kt * (128 if layout == "cdna4" else 64)
            n_lane = nb * rows + lanes % rows
            s_lane, c_lane = s[n_lane, k0 // 128], z[n_lane, k0 // 128]
            for a in range(4):
                got = _blocks_dequant(buf[nb, kt, :, a], s_lane, c_lane)   # (the kernels fold the magic / offset: q s + sz exactly)
                kcol = k0 + kspan * a + 8 * (lanes // rows)
                want = np.stack([W[n_lane, kcol + j] for j in range(8)], axis=1)
                assert np.allclose(got, want, rtol=0, atol=1e-12), (layout, nb, kt, a)


def test_product_mfma_32x32x16_on_the_emitted_operands():
    """the emitted operands against activations in the B layout of v_mfma_f32_32x32x16 (lane (m = l % 32, kb = l / 32): 8 k of row m), D
    in its accumulator layout (lane l, register r: weight row (r & 3) + 8 (r >> 2) + 4 (l / 32), activation row l % 32 -- the layout
    awq_gemm_v4.hip's epilogue reads) == W x^T."""
    N, K, M = 32, 128, 32
    rng = np.random.default_rng(11)
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    s = rng.uniform(0.004, 0.01, size=(N, 1))
    z = -s * rng.integers(0, 16, size=(N, 1))
    W = q * s + z
    x = rng.standard_normal((M, K))
    buf = np.ascontiguousarray(O.pack_cdna4w(q)).view(np.uint32).reshape(1, K // 64, 64, 4)
    lanes = np.arange(64)
    acc = np.zeros((64, 16))
    for kt in range(K // 64):
        for a in range(4):
            A_l = _blocks_dequant(buf[0, kt, :, a], s[lanes % 32, 0], z[lanes % 32, 0])       # [lane][8]
            kcol = 64 * kt + 16 * a + 8 * (lanes // 32)
            B_l = np.stack([x[lanes % 32, kcol + j] for j in range(8)], axis=1)            # [lane][8]
            Am = np.zeros((32, 16))
            Bm = np.zeros((16, 32))
            for lane in range(64):
                Am[lane % 32, 8 * (lane // 32): 8 * (lane // 32) + 8] = A_l[lane]
                Bm[8 * (lane // 32): 8 * (lane // 32) + 8, lane % 32] = B_l[lane]
            D = Am @ Bm
            for lane in range(64):
                for r in range(16):
                    acc[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane // 32), lane % 32]
    ref = W @ x.T                                                                            # [n, m]
    for lane in range(64):
        for r in range(16):
            assert abs(acc[lane, r] - ref[(r & 3) + 8 * (r >> 2) + 4 * (lane // 32), lane % 32]) < 1e-9


def test_index_model_of_the_v6w_experiment_kernel():
    """The address arithmetic of csrc/awq_gemm_v6w.hip (AWQ_PROBES builds; not yet run on hardware) restated formula by formula -- x staging
    writes (wpat), fragment reads (xa + f * 8192), weight / scale offsets (w_off, s_off, tile and word of a step), the operand the dequant
    emits, the 32x32x16 product in its register layouts, the epilogue's staging writes and row-major read-back -- on one 256 x 256 block:
    out == x W^T.  Scheduling and hazards are not modelled; a slip in an index expression shows up here before the kernel meets a GPU."""
    M, N, K = 256, 256, 256
    nit = K // 128
    rng = np.random.default_rng(3)
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    s = rng.uniform(0.004, 0.01, size=(N, nit))
    z = -s * rng.integers(0, 16, size=(N, nit))
    W = q * np.repeat(s, 128, axis=1) + np.repeat(z, 128, axis=1)
    x = rng.standard_normal((M, K))
    qw = np.ascontiguousarray(O.pack_cdna4w(q)).view(np.uint32).reshape(-1)             # flat u32 words
    # sz_packed as the kernels read it: [N / 16][K / 128][16] -> here two float arrays with the same indexing
    szs = s.reshape(N // 16, 16, nit).transpose(0, 2, 1).reshape(-1)
    szz = z.reshape(N // 16, 16, nit).transpose(0, 2, 1).reshape(-1)
    m0 = n0 = 0
    pitch = 2 * 256 + 16
    stage_out = np.zeros((256 * pitch // 2,))                                             # epilogue staging, in units of one T element
    out = np.zeros((M, N))
    lanes = np.arange(64)
    l32, kb = lanes & 31, lanes >> 5
    r4, p16 = lanes >> 4, lanes & 15
    accs = {}
    for wv in range(4):
        acc = np.zeros((8, 2, 64, 16))
        for t in range(nit):
            # ---- x tile t in LDS as the staging writes lay it out (granule units of 8 elements) ----
            lds = np.zeros((256 * 16, 8))                                                 # [row * 16 + slot][8 k]
            for w2 in range(4):                                                           # all four waves stage their 64 rows
                for qq in range(16):
                    for ln in range(64):
                        row_src = m0 + 64 * w2 + 4 * qq + r4[ln]                          # xw + 4 q K + r4 K
                        kcol = t * 128 + p16[ln] * 8                                      # kt * 128 + p16 * 8
                        wpat = (64 * w2) * 256 + r4[ln] * 256 + ((p16[ln] ^ (4 * (qq & 3) + r4[ln])) << 4)
                        addr = wpat + qq * 1024                                           # bytes inside the stage
                        lds[addr // 16] = x[row_src, kcol: kcol + 8]
            for S in range(8):
                for p in range(2):
                    pc = (n0 >> 5) + 2 * wv + p
                    TI, WI = S >> 2, S & 3
                    w_off = pc * (2 * nit) * 256 + lanes * 4
                    words = qw[(2 * t + TI) * 256 + w_off + WI]
                    slab = 2 * pc + (l32 >> 4)
                    s_off = slab * nit * 16 + (lanes & 15)
                    A_l = _blocks_dequant(words, szs[t * 16 + s_off], szz[t * 16 + s_off])   # [lane][8]
                    for f in range(8):
                        xa = l32 * 256 + (((2 * S + kb) ^ (lanes & 15)) << 4)
                        B_l = lds[(xa + f * 8192) // 16]                                  # [lane][8]
                        Am = np.zeros((32, 16))
                        Bm = np.zeros((16, 32))
                        for ln in range(64):
                            Am[ln % 32, 8 * (ln // 32): 8 * (ln // 32) + 8] = A_l[ln]
                            Bm[8 * (ln // 32): 8 * (ln // 32) + 8, ln % 32] = B_l[ln]
                        D = Am @ Bm
                        for ln in range(64):
                            for r in range(16):
                                acc[f, p, ln, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (ln // 32), ln % 32]
        accs[wv] = acc
    # ---- epilogue: staging writes of every wave, then the row-major read-back / store ----
    for wv in range(4):
        for f in range(8):
            for p in range(2):
                for rg in range(4):
                    for ln in range(64):
                        wbase = l32[ln] * pitch + (64 * wv + 4 * kb[ln]) * 2
                        addr = wbase + f * (32 * pitch) + p * 64 + rg * 16                # bytes; four consecutive T elements
                        for e in range(4):
                            stage_out[addr // 2 + e] = accs[wv][f, p, ln, 4 * rg + e]
    for wv in range(4):
        for it in range(32):
            for ln in range(64):
                col = (ln & 31) * 8
                row = 64 * wv + 2 * it + (ln >> 5)
                ra = row * pitch + col * 2
                out[m0 + row, n0 + col: n0 + col + 8] = stage_out[ra // 2: ra // 2 + 8]
    ref = x @ W.T
    assert np.allclose(out, ref, rtol=0, atol=1e-8), np.abs(out - ref).max()


def test_decode_kernels_can_read_cdna4w_at_slab_granularity():
    """A 16-row slab is HALF of every pair tile (tile lanes 32 kb + 16 h + 0..15), so a decode block per slab -- the granularity that
    gives N = 4096 projections one block per CU -- fetches, per 128-k step, the two halves of two adjacent pair tiles with ONE wave load
    (per-lane offset tsel * 1024 + (32 kb + 16 h + l % 16) * 16, kb = (l / 16) & 1, tsel = l / 32): the 4-lane blocks of the dequant MFMA
    see the same lane quads as inside the tile, and the result is the 16x16x32 A operand of row 16 h + l % 16 over the k set
    {64 (2 st + tsel) + 16 a + 8 kb + 0..7}; with the activations read at the matching offsets the product is W x^T.  So the planned layout
    costs the decode kernels three constants (lane offset of the weight load, lane offset of the x operand, nothing for sz), not their
    block granularity."""
    N, K, M = 64, 256, 5
    rng = np.random.default_rng(19)
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    s = rng.uniform(0.004, 0.01, size=(N, K // 128))
    z = -s * rng.integers(0, 16, size=(N, K // 128))
    W = q * np.repeat(s, 128, axis=1) + np.repeat(z, 128, axis=1)
    x = rng.standard_normal((M, K))
    buf = np.ascontiguousarray(O.pack_cdna4w(q)).view(np.uint8).reshape(-1)                # bytes, [pair][K / 64] tiles of 1 KiB
    lanes = np.arange(64)
    i16, g = lanes & 15, lanes >> 4
    kbl, tsel = g & 1, g >> 1
    ref = W @ x.T                                                                            # [n, m]
    for slab in range(N // 16):
        pair, h = slab >> 1, slab & 1
        acc = np.zeros((16, 16))                                                            # [n in slab][m]  (16x16x32: D[i][j])
        for st in range(K // 128):                                                          # one 128-k step = one wave load of 1 KiB
            voff = tsel * 1024 + (32 * kbl + 16 * h + i16) * 16                             # per-lane constant
            soff = (pair * (K // 64) + 2 * st) * 1024                                       # wave-uniform: pair tile 2 st (and 2 st + 1 behind it)
            words = np.stack([np.array([int.from_bytes(bytes(buf[soff + v + 4 * a: soff + v + 4 * a + 4]), "little") for v in voff], dtype=np.uint32)
                              for a in range(4)], axis=1)                                   # [lane][word a]: the lane's 16 bytes
            n_lane = 16 * slab + i16
            for a in range(4):
                A_l = _blocks_dequant(words[:, a], s[n_lane, st], z[n_lane, st])            # [lane][8]: row n_lane, k = base + 0..7
                kbase = 64 * (2 * st + tsel) + 16 * a + 8 * kbl
                want = np.stack([W[n_lane, kbase + j] for j in range(8)], axis=1)
                assert np.allclose(A_l, want, rtol=0, atol=1e-12), (slab, st, a)
                # 16x16x32: A lane (i = l % 16, g = l / 16) holds 8 of the 32 k; B lane (j = l % 16, g) the same k of activation row j
                B_l = np.stack([x[np.minimum(i16, M - 1), kbase + j] for j in range(8)], axis=1)
                for gg in range(4):
                    Ag = A_l[16 * gg: 16 * gg + 16]                                         # [i][8]
                    Bg = B_l[16 * gg: 16 * gg + 16]                                         # [j][8]
                    acc += Ag @ Bg.T
        assert np.allclose(acc[:, :M], ref[16 * slab: 16 * slab + 16, :], rtol=0, atol=1e-9), slab


def test_forward_rule_of_the_device_repacker_inverts_the_index_map():
    """csrc/awq_gemm_v6w.hip::repack_v2_to_cdna4w_kernel builds destination word t = tile * 256 + lane * 4 + a nibble by nibble from
    (n, k) = (32 pair + 4 nq + 2 (i & 1) + hi, 64 kt + 16 a + 8 kb + 4 (i >> 1) + r): that rule, restated, is the inverse of
    cdna4w_position for every (word, nibble)."""
    N, K = 64, 256
    nt64 = K // 64
    for t in range(N * K // 8):
        a, lane, tile = t & 3, (t >> 2) & 63, t >> 8
        pair, kt = tile // nt64, tile % nt64
        kb, nq, r = lane >> 5, (lane >> 2) & 7, lane & 3
        for p in range(8):
            i, hi = p & 3, p >> 2
            n = 32 * pair + 4 * nq + 2 * (i & 1) + hi
            k = 64 * kt + 16 * a + 8 * kb + 4 * (i >> 1) + r
            word, pp = O.cdna4w_position(n, k, K)
            assert (int(word), int(pp)) == (t, p), (t, p, n, k)
