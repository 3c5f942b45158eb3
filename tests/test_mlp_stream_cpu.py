"""Host-side check of the fused message-MLP kernel's weight stream (csrc/mlp_fused.hip, og_mlp_block_pack) WITHOUT a GPU: a numpy
emulation of the kernel's data flow -- the stage order, the MFMA operand / accumulator lane layouts the HIP kernels rely on
(og_common.h: mfma32_row; cdna_hip_programming.md §3) and the re-use of fc.0's accumulator registers as fc.3's B fragments --
consumes the packed stream exactly as the kernel does and must reproduce x + W3 relu(W0 [x ; O] + b0) + b3
(reference attention_gnn.py:53-55 + models/utils.py:48-58 after the folds of og_pack_weights)."""
import numpy as np
import pytest
import torch

from openglue_amd import _lib


def _mfma_32x32x16(a_frag, b_frag, acc):
    """v_mfma_f32_32x32x16_f16 on lane-level fragments: a_frag, b_frag [64 lanes][8], acc [64 lanes][16] (float64 here).
    A[row = l & 31][k = 8 (l >> 5) + e], B[k = 8 (l >> 5) + e][col = l & 31], D lane l reg r = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
        Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
    Dm = A @ Bm
    for l in range(64):
        for r in range(16):
            acc[l, r] += Dm[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]


def _split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


@pytest.mark.parametrize("D", [256, 128])
def test_mlp_stream_numpy_emulation_of_the_kernel(D):
    """D = 256: two hidden halves x two quarters of 4 hidden blocks, fc.3 stages (hidden block j, k-step t) over 8 output blocks.
    D = 128 (round 5; the reference's SIFT / HardNet width, attention_gnn.py:41 is generic in D): two halves x ONE pass of 4 hidden blocks,
    fc.3 stages = hidden block j with both k-steps over 4 output blocks (mlp_fused_kernel<128>)."""
    lib = _lib.load()
    nbytes = lib.og_mlp_block_stream_bytes(D)
    assert nbytes == 6 * D * D * 4
    assert lib.og_mlp_block_stream_bytes(64) == 0 and lib.og_mlp_block_pack(64, None, None, None) != 0
    G0, NOB = 2 * D // 32, D // 32
    HB2 = G0 // 2; NPASS = HB2 // 4; NJT = 8 if D == 256 else 4; SPP = G0 + NJT
    g = torch.Generator().manual_seed(11)
    w0 = (torch.randn(2 * D, 2 * D, generator=g) * 0.04).contiguous()
    w3 = (torch.randn(D, 2 * D, generator=g) * 0.05).contiguous()
    b0 = (torch.randn(2 * D, generator=g) * 0.3).double().numpy()
    b3 = (torch.randn(D, generator=g) * 0.3).double().numpy()
    st = torch.empty(nbytes, dtype=torch.uint8)
    assert lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), st.data_ptr()) == 0
    halves = st.numpy().view(np.float16).astype(np.float64).reshape(NPASS * SPP, 32, 64, 8)      # [stage][fragment][lane][element]

    xo = (torch.randn(32, 2 * D, generator=g) * 1.5).double().numpy()                   # one wave: 32 tokens of [x | O]
    xh, xl = _split(xo)
    xo_rep = xh + xl
    lanes = np.arange(64)
    tok, hh = lanes & 31, lanes >> 5

    # one token block = two waves: wave a owns hidden half a and a PARTIAL fc.3 sum over all output blocks
    acc3 = np.zeros((2, NOB, 64, 16))
    for i in range(NOB):                                     # the bias enters once: in the wave that finishes the block
        for l in range(64):
            for r in range(16):
                acc3[i // (NOB // 2), i, l, r] = 256.0 * b3[32 * i + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]
    for a in range(2):
        for q in range(NPASS):
            acc0 = np.zeros((4, 64, 16))
            for i in range(4):
                for l in range(64):
                    for r in range(16):
                        acc0[i, l, r] = 256.0 * b0[32 * (HB2 * a + 4 * q + i) + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]
            for kg in range(G0):                             # fc.0 stages
                s = SPP * q + kg
                for t in range(2):
                    cols = (32 * kg + 16 * t + 8 * hh)[:, None] + np.arange(8)[None, :]
                    bh, bl = xh[tok[:, None], cols], xl[tok[:, None], cols]
                    for i in range(4):
                        f = ((a * 2 + t) * 4 + i) * 2
                        wh, wl = halves[s, f], halves[s, f + 1]
                        _mfma_32x32x16(wl, bh, acc0[i]); _mfma_32x32x16(wh, bl, acc0[i]); _mfma_32x32x16(wh, bh, acc0[i])
            for j in range(4):                               # fc.3: hidden block j of this pass from acc0[j]
                v = np.maximum(acc0[j].astype(np.float32) * np.float32(1.0 / 256.0), 0).astype(np.float64)
                vh, vl = _split(v)
                for t in range(2):
                    bh, bl = vh[:, 8 * t:8 * t + 8], vl[:, 8 * t:8 * t + 8]      # element e of k-step t = accumulator register 8t + e
                    for i in range(NOB):
                        if D == 256:
                            s = SPP * q + G0 + 2 * j + t; f = (a * 8 + i) * 2           # stage (j, t), fragment (a, output block i)
                        else:
                            s = SPP * q + G0 + j; f = ((a * 2 + t) * 4 + i) * 2          # stage j, fragment (a, t, output block i)
                        wh, wl = halves[s, f], halves[s, f + 1]
                        _mfma_32x32x16(wl, bh, acc3[a, i]); _mfma_32x32x16(wh, bl, acc3[a, i]); _mfma_32x32x16(wh, bh, acc3[a, i])
    acc3 = acc3[0] + acc3[1]                                 # the exchange at the end of the kernel
    out = np.zeros((32, D))
    for i in range(NOB):
        for l in range(64):
            for r in range(16):
                ch = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                out[l & 31, ch] = acc3[i, l, r] / 256.0 + xo_rep[l & 31, ch]
    ref = xo_rep[:, :D] + np.maximum(xo_rep @ w0.double().numpy().T + b0, 0) @ w3.double().numpy().T + b3
    err = np.abs(out - ref).max()
    print(f"emulated kernel vs float64: {err:.2e}")
    assert err < 2e-5          # fp32 rounding of the hidden activation + the dropped lo*lo terms


@pytest.mark.parametrize("D", [256, 128])
def test_mlp_small_kernel_addresses_the_same_stream(D):
    """mlp_small_kernel (few token rows: 32-token workgroups) reads its weight fragments straight from the big kernel's stream with per-wave
    address formulas (csrc/mlp_fused.hip: frag0 / frag3).  D = 256: wave w owns hidden blocks 2w, 2w+1 in fc.0 and OUTPUT block w over all 16
    hidden blocks in fc.3; D = 128: wave w owns hidden block w in fc.0 and output block w & 3 over hidden blocks 4 (w >> 2) .. + 3 in fc.3 (the two
    halves are added).  Emulated here with numpy, fragment by fragment at exactly those byte offsets: it must reproduce
    x + W3 relu(W0 [x ; O] + b0) + b3."""
    lib = _lib.load()
    G0, NOB = 2 * D // 32, D // 32
    NJ, HB2, SPP = G0 // 8, G0 // 2, G0 + (8 if D == 256 else 4)
    g = torch.Generator().manual_seed(12)
    w0 = (torch.randn(2 * D, 2 * D, generator=g) * 0.04).contiguous()
    w3 = (torch.randn(D, 2 * D, generator=g) * 0.05).contiguous()
    b0 = (torch.randn(2 * D, generator=g) * 0.3).double().numpy()
    b3 = (torch.randn(D, generator=g) * 0.3).double().numpy()
    st = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
    assert lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), st.data_ptr()) == 0
    flat = st.numpy().view(np.float16).astype(np.float64)                     # the stream as halves
    WSTAGE = 32768

    def frag(byte_off):                                                        # one fragment: 1 KiB = [lane][8 halves]
        assert byte_off % 1024 == 0
        return flat[byte_off // 2:byte_off // 2 + 512].reshape(64, 8)

    xo = (torch.randn(32, 2 * D, generator=g) * 1.5).double().numpy()
    xh, xl = _split(xo)
    xo_rep = xh + xl
    lanes = np.arange(64)
    tok, hh = lanes & 31, lanes >> 5
    hid_h = np.zeros((G0, 2, 64, 8)); hid_l = np.zeros((G0, 2, 64, 8))       # the LDS hand-over: [hidden block][t][lane][e]
    for w in range(8):                                                         # fc.0: wave w, hidden blocks NJ w + j
        for j in range(NJ):
            hb = NJ * w + j
            a, q, i = hb // HB2, (hb % HB2) >> 2, hb & 3
            acc = np.zeros((64, 16))
            for l in range(64):
                for r in range(16):
                    acc[l, r] = 256.0 * b0[32 * hb + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]
            for ks in range(2 * G0):
                kg, t = ks >> 1, ks & 1
                cols = (32 * kg + 16 * t + 8 * hh)[:, None] + np.arange(8)[None, :]
                bh, bl = xh[tok[:, None], cols], xl[tok[:, None], cols]
                base = (SPP * q + kg) * WSTAGE + ((((a * 2 + t) * 4 + i) * 2) << 10)          # frag0(dst, j, kg, t, part)
                wh, wl = frag(base), frag(base + 1024)
                _mfma_32x32x16(wl, bh, acc); _mfma_32x32x16(wh, bl, acc); _mfma_32x32x16(wh, bh, acc)
            v = np.maximum(acc.astype(np.float32) * np.float32(1.0 / 256.0), 0).astype(np.float64)
            vh, vl = _split(v)
            for t in range(2):
                hid_h[hb, t], hid_l[hb, t] = vh[:, 8 * t:8 * t + 8], vl[:, 8 * t:8 * t + 8]      # element e of k-step t = accumulator register 8 t + e
    accs = np.zeros((8, 64, 16))
    for w in range(8):                                                         # fc.3: wave w, output block w & (NOB - 1)
        ob = w & (NOB - 1)
        hbs = range(16) if D == 256 else range(4 * (w >> 2), 4 * (w >> 2) + 4)
        for hb in hbs:
            for t in range(2):
                if D == 256:
                    base = (24 * ((hb >> 2) & 1) + 16 + 2 * (hb & 3) + t) * WSTAGE + ((((hb >> 3) * 8 + ob) * 2) << 10)      # frag3(dst, hb, t, part)
                else:
                    base = (8 + (hb & 3)) * WSTAGE + (((((hb >> 2) * 2 + t) * 4 + ob) * 2) << 10)
                wh, wl = frag(base), frag(base + 1024)
                _mfma_32x32x16(wl, hid_h[hb, t], accs[w]); _mfma_32x32x16(wh, hid_l[hb, t], accs[w]); _mfma_32x32x16(wh, hid_h[hb, t], accs[w])
    out = np.zeros((32, D))
    for ob in range(NOB):
        acc = accs[ob] if D == 256 else accs[ob] + accs[ob + 4]                # 128-d: the two hidden halves meet in LDS
        for l in range(64):
            for r in range(16):
                ch = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                out[l & 31, ch] = acc[l, r] / 256.0 + b3[ch] + xo_rep[l & 31, ch]
    ref = xo_rep[:, :D] + np.maximum(xo_rep @ w0.double().numpy().T + b0, 0) @ w3.double().numpy().T + b3
    err = np.abs(out - ref).max()
    print(f"emulated small kernel vs float64: {err:.2e}")
    assert err < 2e-5


def test_mlp_stream_range_error():
    lib = _lib.load()
    D = 256
    w0 = torch.zeros(2 * D, 2 * D); w3 = torch.zeros(D, 2 * D)
    w0[3, 5] = 300.0                                  # 256 * 300 > 65504
    st = torch.empty(lib.og_mlp_block_stream_bytes(D), dtype=torch.uint8)
    assert lib.og_mlp_block_pack(D, w0.data_ptr(), w3.data_ptr(), st.data_ptr()) == -5


@pytest.mark.parametrize("K", [256, 128])
def test_proj_stream_layout(K):
    """og_proj_block_pack (the small-batch projection kernel's weight stream): fragment ((i * K / 16 + ks) * 2 + part) of 1 KiB, lane l =
    (rho = l & 31, h = l >> 5), element e = 256 w[32 i + rho][16 ks + 8 h + e] as (hi, lo) halves -- re-read here with numpy."""
    lib = _lib.load()
    N = 96
    assert lib.og_proj_block_stream_bytes(N, K) == N * K * 4          # (N not a multiple of 128: the small-batch stream only)
    assert lib.og_proj_block_stream_bytes(100, K) == 0 and lib.og_proj_block_stream_bytes(N, 64) == 0
    g = torch.Generator().manual_seed(5)
    w = torch.randn(N, K, generator=g) * 0.05
    st = torch.empty(N * K * 4, dtype=torch.uint8)
    assert lib.og_proj_block_pack(N, K, w.data_ptr(), st.data_ptr()) == 0
    h = st.numpy().view(np.float16).reshape(N // 32, K // 16, 2, 64, 8).astype(np.float64)       # [i][ks][part][lane][e]
    back = np.zeros((N, K))
    for l in range(64):
        rho, hh = l & 31, l >> 5
        for e in range(8):
            back[rho::32, 8 * hh + e::16] = (h[:, :, 0, l, e] + h[:, :, 1, l, e])
    assert np.abs(back / 256.0 - w.double().numpy()).max() < 2e-8
    assert lib.og_proj_block_pack(N, K, (w * 1e4).contiguous().data_ptr(), st.data_ptr()) == -5      # OG_E_RANGE: 256 w leaves binary16


@pytest.mark.parametrize("K", [256, 128])
def test_proj_stream_big_layout_and_kernel_emulation(K):
    """The BATCH projection kernel's stream (csrc/mlp_fused.hip: og_pack_proj_stream_big, behind the small-batch stream in og_proj_block_pack's
    output when N is a multiple of 128) consumed exactly as proj_stream_kernel does: stage (sp, kq) = sp K/64 + kq, wave half a reads fragments
    [16 a, 16 a + 16), group g = k-step 4 kq + g, blocks 2a + j of super-pair sp; B fragments = the token's (hi, lo) halves in standard k order.
    Must reproduce x W^T (q / k / v projections, attention_gnn.py:43-47)."""
    lib = _lib.load()
    N = 3 * K
    small = N * K * 4
    assert lib.og_proj_block_stream_bytes(N, K) == 2 * small
    g = torch.Generator().manual_seed(9)
    w = (torch.randn(N, K, generator=g) * 0.05).contiguous()
    st = torch.empty(2 * small, dtype=torch.uint8)
    assert lib.og_proj_block_pack(N, K, w.data_ptr(), st.data_ptr()) == 0
    SPS = K // 64
    halves = st.numpy()[small:].view(np.float16).astype(np.float64).reshape(N // 128 * SPS, 32, 64, 8)     # [stage][fragment][lane][element]
    x = (torch.randn(32, K, generator=g) * 1.5).double().numpy()            # one wave: 32 tokens
    xh, xl = _split(x)
    lanes = np.arange(64)
    tok, hh = lanes & 31, lanes >> 5
    out = np.zeros((32, N))
    for sp in range(N // 128):
        for a in range(2):
            acc = np.zeros((2, 64, 16))
            for kq in range(SPS):
                for gq in range(4):
                    ks = 4 * kq + gq
                    cols = (16 * ks + 8 * hh)[:, None] + np.arange(8)[None, :]
                    bh, bl = xh[tok[:, None], cols], xl[tok[:, None], cols]
                    for j in range(2):
                        f = a * 16 + (gq * 2 + j) * 2
                        wh, wl = halves[sp * SPS + kq, f], halves[sp * SPS + kq, f + 1]
                        _mfma_32x32x16(wl, bh, acc[j]); _mfma_32x32x16(wh, bl, acc[j]); _mfma_32x32x16(wh, bh, acc[j])
            for j in range(2):
                for l in range(64):
                    for r in range(16):
                        out[l & 31, 32 * (4 * sp + 2 * a + j) + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)] = acc[j, l, r] / 256.0
    ref = (xh + xl) @ w.double().numpy().T
    err = np.abs(out - ref).max()
    print(f"emulated proj_stream_kernel vs float64: {err:.2e}")
    assert err < 1e-5
