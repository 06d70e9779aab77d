"""Index-level emulation of csrc/token_gemm.hip on the CPU.

The kernel cannot run without a GPU, and its risk is in the index arithmetic (LDS layouts, staging map,
MFMA operand / accumulator layouts, epilogue addressing), not in the arithmetic itself.  This test
transcribes those formulas into numpy, with the 32x32x16 MFMA emulated through the operand / accumulator
layout that attn.hip relies on (validated on hardware by tests/test_attn_gpu.py):
    A operand of lane l : A[row = l & 31][k = 8 (l >> 5) + 0..7]
    B operand of lane l : B[k = 8 (l >> 5) + 0..7][col = l & 31]
    accumulator reg r   : C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
and checks the emulated kernel against x @ W^T + b for ragged T, partial weight blocks and strided x.
If a formula in token_gemm.hip changes, change it here too."""
import numpy as np
import pytest

SLAB_K, SLAB_PAD, WAVES = 64, 72, 4


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag, b_frag [64, 8]; acc [64, 16] -- one v_mfma_f32_32x32x16_bf16 on emulated lanes."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
        Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
    C = A @ Bm
    for l in range(64):
        for r in range(16):
            acc[l, r] += C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]


def emulate(x, ldx, T, w, bias, N, K, NB, relu, grid_x):
    KP, NS = K + 8, K // SLAB_K
    ny = (N + NB * 32 - 1) // (NB * 32)
    y = np.full((T, N), np.nan)
    tiles = (T + 31) // 32
    for by in range(ny):
        n0 = by * NB * 32
        Ws = np.zeros(NB * 32 * KP)
        for c in range(NB * 32 * (K // 8)):                               # weight staging: chunk c of 8 elements
            row, col = c // (K // 8), (c % (K // 8)) * 8
            if n0 + row < N:
                Ws[row * KP + col:row * KP + col + 8] = w[(n0 + row) * K + col:(n0 + row) * K + col + 8]
        bias_s = np.array([bias[n0 + c] if n0 + c < N else 0.0 for c in range(NB * 32)])
        for bx in range(grid_x):
            for wave in range(WAVES):
                wave_id, wave_n = bx * WAVES + wave, grid_x * WAVES
                tile = wave_id
                while tile < tiles:
                    acc = np.zeros((NB, 64, 16))
                    for s in range(NS):
                        slab = np.zeros(32 * SLAB_PAD)
                        for lane in range(64):                            # request() + the slab store
                            for j in range(4):
                                c = lane + 64 * j
                                row, piece = c >> 3, c & 7
                                t = tile * 32 + row
                                v = x[t * ldx + s * SLAB_K + piece * 8:t * ldx + s * SLAB_K + piece * 8 + 8] if t < T else np.zeros(8)
                                slab[row * SLAB_PAD + piece * 8:row * SLAB_PAD + piece * 8 + 8] = v
                        for ks in range(SLAB_K // 16):
                            xb = np.stack([slab[(l & 31) * SLAB_PAD + ks * 16 + (l >> 5) * 8:][:8] for l in range(64)])
                            for b in range(NB):
                                wa = np.stack([Ws[(b * 32 + (l & 31)) * KP + s * SLAB_K + ks * 16 + (l >> 5) * 8:][:8] for l in range(64)])
                                mfma_32x32x16(wa, xb, acc[b])
                    for lane in range(64):                                # epilogue
                        half = lane >> 5
                        t = tile * 32 + (lane & 31)
                        if t >= T:
                            continue
                        for b in range(NB):
                            for g in range(4):
                                nn = b * 32 + 8 * g + 4 * half
                                if n0 + nn < N:
                                    for i in range(4):
                                        v = acc[b, lane, 4 * g + i] + bias_s[nn + i]
                                        y[t, n0 + 4 * half + b * 32 + 8 * g + i] = max(v, 0.0) if relu else v
                    tile += wave_n
    return y


@pytest.mark.parametrize("T,K,N,NB,relu,grid_x", [(70, 128, 64, 8, False, 1), (45, 256, 40, 4, True, 2),
                                                  (33, 128, 264, 8, False, 1), (96, 256, 256, 8, False, 1),
                                                  (40, 512, 136, 4, True, 1)])
def test_token_gemm_index_arithmetic(T, K, N, NB, relu, grid_x):
    rng = np.random.default_rng(T + N)
    ldx = K + 16
    xfull = rng.standard_normal((T, ldx))
    w = rng.standard_normal((N, K)) / np.sqrt(K)
    bias = rng.standard_normal(N)
    ref = xfull[:, :K] @ w.T + bias
    if relu:
        ref = np.maximum(ref, 0.0)
    got = emulate(xfull.reshape(-1), ldx, T, w.reshape(-1), bias, N, K, NB, relu, grid_x)
    assert not np.isnan(got).any(), "some outputs were never written"
    assert np.abs(got - ref).max() < 1e-9
