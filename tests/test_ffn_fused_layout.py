"""Index-level emulation of the fused FFN kernel (ape_amd/csrc/ffn_fused.hip) on the CPU: every lane-level formula the kernel
uses -- MFMA 16x16x32 operand / result layouts, the LDS images of the W1 / W2 chunks (row permutation + XOR swizzle as the
LDS-DMA writes them and as the fragment reads address them), the hidden activations re-used as the B operand of the second
MFMA under a permuted k order, the epilogue's channel ownership -- evaluated in float64 and compared with the plain
x + relu(x W1^T + b1) W2^T + b2.  It validates the LAYOUT DESIGN (the formulas are shared verbatim with the kernel's comments);
the kernel itself is validated by tests/test_ops_gpu.py::test_ffn_fused on the GPU."""
import numpy as np

K_IN, HC, N_OUT = 256, 64, 256


def mfma_16x16x32(a_frag, b_frag, acc):
    """v_mfma_f32_16x16x32_bf16 semantics at lane level.  a_frag[lane] = 8 values A[m = lane & 15][k = 8 * (lane >> 4) + e],
    b_frag[lane] = 8 values B[k = 8 * (lane >> 4) + e][n = lane & 15]; acc[lane][r] = D[m = 4 * (lane >> 4) + r][n = lane & 15]."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for lane in range(64):
        m, g = lane & 15, lane >> 4
        A[m, 8 * g: 8 * g + 8] = a_frag[lane]
        B[8 * g: 8 * g + 8, m] = b_frag[lane]
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        n, g = lane & 15, lane >> 4
        out[lane] += D[4 * g: 4 * g + 4, n]
    return out


# ---- LDS images, as the LDS-DMA writes them (lane-linear destination; the swizzle / permutation is on the SOURCE address)
def w2_channel(rho):
    """LDS row rho = ot * 16 + 4 g + r of the W2 chunk holds output channel 32 (ot >> 1) + 8 g + 4 (ot & 1) + r: after the second MFMA
    a lane (g = lane >> 4) owns, for every pair of output tiles q = ot >> 1, the 8 CONSECUTIVE channels 32 q + 8 g .. + 7 of its
    token -- one store instruction then has the four lanes of a token write 64 contiguous bytes"""
    ot, m = rho >> 4, rho & 15
    return (ot >> 1) * 32 + (m >> 2) * 8 + (ot & 1) * 4 + (m & 3)


def stage_w1(W1, c):
    """W1 chunk image: 64 rows (hidden c*64 + rho) x 32 sixteen-byte chunks (8 bf16 each); instruction i of the image writes
    rows 2i, 2i+1: lane l -> row 2i + (l >> 5), PHYSICAL chunk l & 31, which holds LOGICAL chunk (l & 31) ^ (row & 31)"""
    img = np.zeros((64, 32, 8))
    for i in range(32):
        for lane in range(64):
            rho, qp = 2 * i + (lane >> 5), lane & 31
            q = qp ^ (rho & 31)
            img[rho, qp] = W1[c * HC + rho, q * 8: q * 8 + 8]
    return img


def stage_w2(W2, c):
    """W2 chunk image: 256 rows (permuted output channels) x 8 chunks of 8 hidden; instruction i writes rows 8i .. 8i+7:
    lane l -> row 8i + (l >> 3), physical chunk l & 7 = logical chunk ^ ((row >> 1) & 7)"""
    img = np.zeros((256, 8, 8))
    for i in range(32):
        for lane in range(64):
            rho, qp = 8 * i + (lane >> 3), lane & 7
            q = qp ^ ((rho >> 1) & 7)
            img[rho, qp] = W2[w2_channel(rho), c * HC + q * 8: c * HC + q * 8 + 8]
    return img


def read_w1_frag(img, ht, ks, lane):
    """A operand of the first MFMA: hidden row ht*16 + m, k = ks*32 + 8g .. +8 -> logical chunk ks*4 + g"""
    m, g = lane & 15, lane >> 4
    rho = ht * 16 + m
    return img[rho, (ks * 4 + g) ^ (rho & 31)]


def read_w2_frag(img, ot, kk, lane):
    """A operand of the second MFMA under the permuted k order: e < 4 -> hidden (2kk)*16 + 4g + e, e >= 4 -> (2kk+1)*16 + 4g + e-4:
    two 8-byte halves of chunks kk*4 + (g >> 1) and kk*4 + 2 + (g >> 1), at element offset (g & 1) * 4"""
    m, g = lane & 15, lane >> 4
    rho = ot * 16 + m
    sw = (rho >> 1) & 7
    lo = img[rho, (kk * 4 + (g >> 1)) ^ sw][(g & 1) * 4: (g & 1) * 4 + 4]
    hi = img[rho, (kk * 4 + 2 + (g >> 1)) ^ sw][(g & 1) * 4: (g & 1) * 4 + 4]
    return np.concatenate([lo, hi])


def read_w2_frag_permuted(img, ot, kk, lane):
    """W2P variant: the image was staged from the PRE-PERMUTED W2 (ape_amd.packing.permute_ffn_w2), so the lane's 8 values are
    one 16-byte chunk kk*4 + g"""
    m, g = lane & 15, lane >> 4
    rho = ot * 16 + m
    return img[rho, (kk * 4 + g) ^ ((rho >> 1) & 7)]


def fused_ffn_wave(X32, W1, b1, W2, b2, w2_permuted=False):
    """one wave = 32 tokens: returns Y [32, 256] = x + relu(x W1^T + b1) W2^T + b2 computed the kernel's way"""
    hid = W1.shape[0]
    read_w2 = read_w2_frag
    if w2_permuted:
        import torch
        from ape_amd.packing import permute_ffn_w2
        W2 = permute_ffn_w2(torch.from_numpy(W2)).numpy()
        read_w2 = read_w2_frag_permuted
    # X fragments (B operand of the first MFMA): lane (n = lane & 15 -> token rt*16 + n, g): k = ks*32 + 8g .. +8
    xf = [[np.stack([X32[rt * 16 + (lane & 15), ks * 32 + 8 * (lane >> 4): ks * 32 + 8 * (lane >> 4) + 8] for lane in range(64)])
           for ks in range(8)] for rt in range(2)]
    yacc = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(16)]
    for c in range(hid // HC):
        i1, i2 = stage_w1(W1, c), stage_w2(W2, c)
        hacc = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(4)]
        for ht in range(4):
            for ks in range(8):
                wf = np.stack([read_w1_frag(i1, ht, ks, lane) for lane in range(64)])
                for rt in range(2):
                    hacc[ht][rt] = mfma_16x16x32(wf, xf[rt][ks], hacc[ht][rt])
        # bias + ReLU: hacc[ht][rt][lane][r] = H[token rt*16 + (lane & 15)][hidden c*64 + ht*16 + 4g + r]
        for ht in range(4):
            for rt in range(2):
                for lane in range(64):
                    g = lane >> 4
                    hacc[ht][rt][lane] = np.maximum(hacc[ht][rt][lane] + b1[c * HC + ht * 16 + 4 * g: c * HC + ht * 16 + 4 * g + 4], 0.0)
        # the activations ARE the B operand of the second MFMA (k step kk = hidden tiles 2kk, 2kk+1, permuted inside the step)
        for kk in range(2):
            hb = [np.concatenate([hacc[2 * kk][rt], hacc[2 * kk + 1][rt]], axis=1) for rt in range(2)]
            for ot in range(16):
                wf = np.stack([read_w2(i2, ot, kk, lane) for lane in range(64)])
                for rt in range(2):
                    yacc[ot][rt] = mfma_16x16x32(wf, hb[rt], yacc[ot][rt])
    # epilogue: yacc[ot][rt][lane][r] = Y[token rt*16 + (lane & 15)][channel 32 (ot >> 1) + 8 g + 4 (ot & 1) + r]
    Y = np.zeros((32, N_OUT))
    for ot in range(16):
        for rt in range(2):
            for lane in range(64):
                n, g = lane & 15, lane >> 4
                ch = (ot >> 1) * 32 + g * 8 + (ot & 1) * 4
                Y[rt * 16 + n, ch: ch + 4] = yacc[ot][rt][lane] + b2[ch: ch + 4] + X32[rt * 16 + n, ch: ch + 4]
    return Y


def test_fused_ffn_lane_level_formulas_reproduce_the_ffn():
    rng = np.random.default_rng(0)
    hid = 128                                           # two hidden chunks exercise the accumulation across chunks
    X = rng.standard_normal((32, K_IN))
    W1, b1 = rng.standard_normal((hid, K_IN)) / 16, rng.standard_normal(hid)
    W2, b2 = rng.standard_normal((N_OUT, hid)) / 8, rng.standard_normal(N_OUT)
    ref = X + np.maximum(X @ W1.T + b1, 0) @ W2.T + b2
    got = fused_ffn_wave(X, W1, b1, W2, b2)
    assert np.abs(got - ref).max() < 1e-9
    got = fused_ffn_wave(X, W1, b1, W2, b2, w2_permuted=True)          # one ds_read_b128 per W2 fragment on the pre-permuted matrix
    assert np.abs(got - ref).max() < 1e-9


def test_lds_images_are_permutations_without_bank_group_collisions():
    """every (row, logical chunk) lands in exactly one physical slot, and the 8 lanes of an LDS pass (consecutive lanes read
    consecutive rows of one chunk column) hit 8 different 16-byte bank groups"""
    for rho in range(64):
        assert sorted((q ^ (rho & 31)) for q in range(32)) == list(range(32))
    for rho in range(256):
        assert sorted((q ^ ((rho >> 1) & 7)) for q in range(8)) == list(range(8))
    assert sorted(w2_channel(r) for r in range(256)) == list(range(256))
    for q in range(32):                                   # W1 fragment reads: lanes m = 0..7 / 8..15 of one g
        for base in (0, 8):
            assert len({((q ^ ((base + m) & 31)) & 7) for m in range(8)}) == 8
