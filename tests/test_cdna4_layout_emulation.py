"""CPU: a REGISTER-LEVEL emulation of the matrix-core dequant on the shipped "cdna4" tile (oracle/awq_oracle.py::pack_cdna4).

The instruction sequence the kernels run on a tile word -- extractions (w >> 4 i) & 0x000F000F, v_mfma_f32_4x4x4_16B with the per-block
diagonal B operand, pack of the two results -- is restated on numpy and held against the layout's index map: lane (n = l % 16, g = l / 16)
ends up with k = 32 a + 8 g + 0..7 of row n, the A operand of v_mfma_f32_16x16x32 (what the GPU tests confirm on hardware).  (Round 3 also
carried a 32-row sibling layout, "cdna4w", for v_mfma_f32_32x32x16 products; it ran in round 4, tied / lost against the shipped loop and was
removed: profiles/r04_v6w.txt.)
"""
import numpy as np

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


def test_matrix_core_dequant_emits_the_product_operand():
    N, K = 64, 256
    rng = np.random.default_rng(7)
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    s = rng.uniform(0.004, 0.01, size=(N, K // 128))
    z = -s * rng.integers(0, 16, size=(N, K // 128))
    W = q * np.repeat(s, 128, axis=1) + np.repeat(z, 128, axis=1)          # [N, K] exact in float64
    buf = np.ascontiguousarray(O.pack_cdna4(q)).view(np.uint32).reshape(N // 16, K // 128, 64, 4)
    rows, kspan = 16, 32      # operand: lane (n = l % 16, g = l / 16), word a covers k = 32 a + 8 g + 0..7
    lanes = np.arange(64)
    for nb in range(buf.shape[0]):
        for kt in range(buf.shape[1]):
            k0 = kt * 128
            n_lane = nb * rows + lanes % rows
            s_lane, c_lane = s[n_lane, k0 // 128], z[n_lane, k0 // 128]
            for a in range(4):
                got = _blocks_dequant(buf[nb, kt, :, a], s_lane, c_lane)   # (the kernels fold the magic / offset: q s + sz exactly)
                kcol = k0 + kspan * a + 8 * (lanes // rows)
                want = np.stack([W[n_lane, kcol + j] for j in range(8)], axis=1)
                assert np.allclose(got, want, rtol=0, atol=1e-12), (nb, kt, a)
