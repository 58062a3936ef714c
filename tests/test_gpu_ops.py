"""GPU: operator-level parity of the HIP kernels, called through the C ABI (wm_op_*), against plain
torch fp32 references evaluated on the SAME 16-bit-rounded operands.  Tolerances are stated per test:
MFMA accumulates in fp32, so the only differences are summation order and the 16-bit output rounding.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F16 = 0, 1


def _lib():
    from hunyuanworld_mirror_amd import _lib
    return _lib.lib()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _t16(x, dt):
    return x.to(torch.bfloat16 if dt == BF16 else torch.float16)


def _from16(buf, dt):
    return buf.view(torch.bfloat16 if dt == BF16 else torch.float16).float()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 384, 640), (2752, 1024, 1024), (77, 64, 64), (1376, 4096, 1024)])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_gemm_epilogues(dev, M, N, K, dt):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    A = _t16(torch.randn(M, K, generator=g), dt).to(dev)
    W = _t16(torch.randn(N, K, generator=g) / math.sqrt(K), dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    gamma = torch.randn(N, generator=g).to(dev)
    ref = A.float() @ W.float().t() + bias
    L = _lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # EPI 0: f32
    out = torch.empty(M, N, device=dev)
    assert L.wm_op_gemm(dt, 0, _p(A), _p(W), _p(out), _p(bias), None, M, N, K, s) == 0
    torch.cuda.synchronize()
    e = _rel(out, ref)
    print(f"gemm f32 {M}x{N}x{K} dt{dt}: {e:.2e}")
    assert e < 2e-5
    # EPI 1: 16-bit out
    o16 = torch.empty(M, N, device=dev, dtype=torch.int16)
    assert L.wm_op_gemm(dt, 1, _p(A), _p(W), _p(o16), _p(bias), None, M, N, K, s) == 0
    torch.cuda.synchronize()
    assert _rel(_from16(o16, dt), ref) < (6e-3 if dt == BF16 else 8e-4)
    # EPI 2: GELU(erf) 16-bit
    assert L.wm_op_gemm(dt, 2, _p(A), _p(W), _p(o16), _p(bias), None, M, N, K, s) == 0
    torch.cuda.synchronize()
    assert _rel(_from16(o16, dt), torch.nn.functional.gelu(ref)) < (6e-3 if dt == BF16 else 8e-4)
    # EPI 3: X += gamma * (acc + bias)
    X0 = torch.randn(M, N, generator=g).to(dev)
    X = X0.clone()
    assert L.wm_op_gemm(dt, 3, _p(A), _p(W), _p(X), _p(bias), _p(gamma), M, N, K, s) == 0
    torch.cuda.synchronize()
    assert _rel(X, X0 + gamma * ref) < 2e-5


@pytest.mark.parametrize("M,N,K", [(11008, 1024, 1024), (11008, 3072, 256), (2752, 1024, 512), (1376 * 3 + 5, 512, 256), (200, 256, 128)])
@pytest.mark.parametrize("cfg", [4, 5])
def test_gemm_row_band_schedule_is_bit_identical(dev, M, N, K, cfg):
    """gemm_pp2_kernel with the M rows cut into bands of unequal height (wm_launch_gemm's schedule, or a forced band count) must
    give the same bits as the full-height grid: a band only changes WHICH block owns a row, not the order its K sum runs in."""
    g = torch.Generator(device="cpu").manual_seed(M + N + cfg)
    A = _t16(torch.randn(M, K, generator=g), BF16).to(dev)
    W = _t16(torch.randn(N, K, generator=g) / math.sqrt(K), BF16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    gamma = torch.randn(N, generator=g).to(dev)
    X0 = torch.randn(M, N, generator=g).to(dev)
    L = _lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    U, full = (M + 15) // 16, 16 if cfg == 4 else 12
    bmin = (U + full - 1) // full
    forced = sorted({bmin, bmin + 1, min(U, bmin * 2 - 1), min(U, (bmin * 3) // 2 + 1)})

    def run(sched):
        assert L.wm_set_tuning(b"gemm_cfg", cfg) == 0 and L.wm_set_tuning(b"gemm_sched", sched) == 0
        try:
            f32 = torch.empty(M, N, device=dev)
            o16 = torch.empty(M, N, device=dev, dtype=torch.int16)
            ge = torch.empty(M, N, device=dev, dtype=torch.int16)
            X = X0.clone()
            assert L.wm_op_gemm(BF16, 0, _p(A), _p(W), _p(f32), _p(bias), None, M, N, K, s) == 0
            assert L.wm_op_gemm(BF16, 1, _p(A), _p(W), _p(o16), _p(bias), None, M, N, K, s) == 0
            assert L.wm_op_gemm(BF16, 2, _p(A), _p(W), _p(ge), _p(bias), None, M, N, K, s) == 0
            assert L.wm_op_gemm(BF16, 3, _p(A), _p(W), _p(X), _p(bias), _p(gamma), M, N, K, s) == 0
            torch.cuda.synchronize()
            return f32, o16, ge, X
        finally:
            L.wm_set_tuning(b"gemm_cfg", -1); L.wm_set_tuning(b"gemm_sched", -1)

    base = run(0)   # full-height tiles
    assert _rel(base[0], A.float() @ W.float().t() + bias) < 2e-5
    for sched in [-1] + forced:
        got = run(sched)
        for name, a, b in zip(("f32", "t16", "gelu", "resid"), got, base):
            assert torch.equal(a, b), f"{name} differs with gemm_sched={sched} (M={M}, cfg={cfg})"


@pytest.mark.parametrize("K", [64, 128, 192, 256, 448, 1024, 4096])
@pytest.mark.parametrize("cfg", [4, 5])
def test_gemm_pingpong_v3_matches_v2_bitwise(dev, K, cfg):
    """The half-the-barriers schedule (gemm_pp2_kernel VER = 3, tuning gemm_pp = 4) against the shipped two-barriers-per-stage one:
    same MFMAs in the same order, so any difference is a synchronisation fault (a fragment read before its LDS-DMA landed, or a
    region refilled while a wave still read it).  K from one K-tile up (the prologue / tail branches of the counted waits), both
    tile heights, odd row counts, several launches back to back."""
    L = _lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, N in ((11008, 1024), (5000, 768), (1376 * 2 + 3, 256)):
        g = torch.Generator(device="cpu").manual_seed(M + K + cfg)
        A = _t16(torch.randn(M, K, generator=g), BF16).to(dev)
        W = _t16(torch.randn(N, K, generator=g) / math.sqrt(K), BF16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        outs = {}
        for ver in (2, 4):
            assert L.wm_set_tuning(b"gemm_cfg", cfg) == 0 and L.wm_set_tuning(b"gemm_pp", ver) == 0
            try:
                res = []
                for rep in range(4 if ver == 4 else 1):
                    f32 = torch.empty(M, N, device=dev)
                    ge = torch.empty(M, N, device=dev, dtype=torch.int16)
                    assert L.wm_op_gemm(BF16, 0, _p(A), _p(W), _p(f32), _p(bias), None, M, N, K, s) == 0
                    assert L.wm_op_gemm(BF16, 2, _p(A), _p(W), _p(ge), _p(bias), None, M, N, K, s) == 0
                    res.append((f32, ge))
                torch.cuda.synchronize()
                outs[ver] = res
            finally:
                L.wm_set_tuning(b"gemm_cfg", -1); L.wm_set_tuning(b"gemm_pp", -1)
        ref32, refge = outs[2][0]
        assert _rel(ref32, A.float() @ W.float().t() + bias) < 2e-5
        for rep, (f32, ge) in enumerate(outs[4]):
            assert torch.equal(f32, ref32), f"f32 differs: M={M} N={N} K={K} cfg={cfg} rep={rep}"
            assert torch.equal(ge, refge), f"gelu differs: M={M} N={N} K={K} cfg={cfg} rep={rep}"


@pytest.mark.parametrize("K", [64, 128, 192, 1024, 4096])
@pytest.mark.parametrize("cfg", [4, 5])
def test_gemm_residual_prefetch_is_bit_identical(dev, K, cfg):
    """Tuning resid_prefetch = 1: the residual epilogue's old C tile is touched line by line during the last two K-tiles (4-byte LDS-DMA
    into a scratch corner, counted by the same vmcnt as the operand refills — gemm.hip prefetch_c).  It moves no value: X must come out
    bit-identical, from one K-tile (no prefetch window) up, on both tile heights, ragged row counts (rows past M clamp), scheduled bands."""
    L = _lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, N in ((11008, 1024), (5000, 768), (1376 * 2 + 3, 256)):
        g = torch.Generator(device="cpu").manual_seed(M + K + cfg)
        A = _t16(torch.randn(M, K, generator=g), BF16).to(dev)
        W = _t16(torch.randn(N, K, generator=g) / math.sqrt(K), BF16).to(dev)
        bias = torch.randn(N, generator=g).to(dev); gamma = torch.randn(N, generator=g).to(dev)
        X0 = torch.randn(M, N, generator=g).to(dev)
        outs = {}
        for pf in (0, 1):
            assert L.wm_set_tuning(b"gemm_cfg", cfg) == 0 and L.wm_set_tuning(b"resid_prefetch", pf) == 0
            try:
                res = []
                for rep in range(3 if pf else 1):
                    X = X0.clone()
                    assert L.wm_op_gemm(BF16, 3, _p(A), _p(W), _p(X), _p(bias), _p(gamma), M, N, K, s) == 0
                    res.append(X)
                torch.cuda.synchronize()
                outs[pf] = res
            finally:
                L.wm_set_tuning(b"gemm_cfg", -1); L.wm_set_tuning(b"resid_prefetch", -1)
        ref = outs[0][0]
        assert _rel(ref, X0 + gamma * (A.float() @ W.float().t() + bias)) < 2e-5
        for rep, X in enumerate(outs[1]):
            assert torch.equal(X, ref), f"M={M} N={N} K={K} cfg={cfg} rep={rep}"


@pytest.mark.parametrize("M,K", [(11008, 1024), (11008, 4096), (10992, 1024), (2752, 512), (1376 * 5 + 3, 256), (12288, 128)])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_gemm_residual_with_fused_layernorm(dev, M, K, dt):
    """X += gamma (A W^T + b) with the following LayerNorm fused into the epilogue (the four column tiles of a row band exchange per-row
    (mean, M2) partials inside the launch; gemm.hip epilogue_resid_ln).  Checked: X bit-identical to the unfused epilogue; the LayerNorm
    output against fp32 torch LayerNorm of THAT X at the 16-bit tolerance, and against the library's own LayerNorm kernel up to rounding
    flips (partial-combined statistics instead of a two-pass row: a few outputs differ by one ulp); several launches back to back on the
    same buffers (the counters must come back to zero), ragged row counts, both tile schedules."""
    N = 1024
    g = torch.Generator(device="cpu").manual_seed(M + K + dt)
    A = _t16(torch.randn(M, K, generator=g), dt).to(dev)
    W = _t16(torch.randn(N, K, generator=g) / math.sqrt(K), dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev); gamma = (torch.randn(N, generator=g) * 0.3).to(dev)
    lw = torch.randn(N, generator=g).to(dev); lb = torch.randn(N, generator=g).to(dev)
    X0 = (torch.randn(M, N, generator=g) * 2 + 0.5).to(dev)
    X0[:, 7] += 40.0    # a massive-activation channel: the row statistics must not lose the rest against it
    L = _lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Xref = X0.clone()
    assert L.wm_op_gemm(dt, 3, _p(A), _p(W), _p(Xref), _p(bias), _p(gamma), M, N, K, s) == 0
    ln_ref16 = torch.empty(M, N, device=dev, dtype=torch.int16)
    assert L.wm_op_layernorm(_p(Xref), _p(ln_ref16), _p(lw), _p(lb), M, N, 1e-5, 0, dt, s) == 0
    stats = torch.empty(M * 8, device=dev); sync = torch.zeros(3 * (M // 16 + 2), device=dev, dtype=torch.int32)
    fused = C.c_int(-1)
    for sched in (-1, 0):
        assert L.wm_set_tuning(b"gemm_sched", sched) == 0 and L.wm_set_tuning(b"ln_fuse", 1) == 0   # (the fused epilogue is opt-in: gemm.hip wm_gemm_fuses_ln)
        try:
            for rep in range(3):
                X = X0.clone()
                out16 = torch.full((M, N), 0x7FC0 if dt == BF16 else 0x7E00, device=dev, dtype=torch.int16)
                assert L.wm_op_gemm_resid_ln(dt, _p(A), _p(W), _p(X), _p(bias), _p(gamma), _p(lw), _p(lb), 1e-5, _p(out16), _p(stats), _p(sync), M, N, K,
                                             C.byref(fused), s) == 0
                torch.cuda.synchronize()
                assert torch.equal(X, Xref), "residual stream differs from the unfused epilogue"
                if fused.value != 1:      # shapes the launcher keeps on another tile configuration: X only, ln_out untouched
                    assert M < 10000, "the 8-view backbone shapes must take the fused epilogue"
                    e, diff = float("nan"), float("nan")
                    continue
                assert int(sync.abs().sum()) == 0, "rendezvous counters / fallback flags not back at zero"
                got = _from16(out16, dt)
                ref = torch.nn.functional.layer_norm(Xref, (N,), lw, lb, 1e-5)
                e = _rel(got, ref)
                diff = float((out16 != ln_ref16).float().mean())
                assert torch.isfinite(got).all() and e < (4e-3 if dt == BF16 else 6e-4), e
                assert diff < 0.01, f"{100 * diff:.2f} % of the outputs differ from the LayerNorm kernel's"
        finally:
            L.wm_set_tuning(b"gemm_sched", -1); L.wm_set_tuning(b"ln_fuse", -1)
    print(f"fused LN M{M} K{K} dt{dt}: rel-L2 vs torch {e:.2e}, {100 * diff:.3f} % of the 16-bit outputs differ from the LayerNorm kernel")


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric W catches a transposed C-write (guides §3)."""
    K = N = 128
    A = torch.eye(K).to(torch.bfloat16).to(dev)
    W = (torch.arange(N)[:, None] * 2.0 + torch.arange(K)[None, :] * 0.25).to(torch.bfloat16).to(dev)
    out = torch.empty(K, N, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_gemm(BF16, 0, _p(A), _p(W), _p(out), None, None, K, N, K, s) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, W.float().t())


LOG2E = 1.4426950408889634


def _attn_ref(q, k, v):
    # q,k,v [H, L, 64] fp32; q carries log2(e)/sqrt(d) (the kernel evaluates softmax in base 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * math.log(2.0), -1)
    return a @ v


@pytest.mark.parametrize("H,nseq,L", [(16, 2, 1376), (16, 3, 1374), (2, 3, 27), (4, 1, 2752), (2, 1, 64), (2, 2, 100)])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_attention(dev, H, nseq, L, dt):
    g = torch.Generator().manual_seed(H * 1000 + L)
    M = nseq * L
    q = _t16(torch.randn(H, M, 64, generator=g) * 0.125 * 1.5 * LOG2E, dt).to(dev)
    k = _t16(torch.randn(H, M, 64, generator=g) * 1.5, dt).to(dev)
    v = _t16(torch.randn(H, M, 64, generator=g), dt).to(dev)
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_attention(dt, _p(q), _p(k), _p(v), _p(o), H, M, L, 1, 0, s) == 0
    torch.cuda.synchronize()
    got = _from16(o, dt).reshape(M, H, 64)
    ref = torch.empty(M, H, 64, device=dev)
    for i in range(nseq):
        sl = slice(i * L, (i + 1) * L)
        ref[sl] = _attn_ref(q[:, sl].float(), k[:, sl].float(), v[:, sl].float()).transpose(0, 1)
    e = _rel(got, ref)
    print(f"attention H{H} nseq{nseq} L{L} dt{dt}: {e:.2e}")
    # P is rounded to 16 bit before P@V and O is stored in 16 bit
    assert e < (8e-3 if dt == BF16 else 1.5e-3)


def test_attention_spike_forces_rescale(dev):
    """Online-softmax rescale branch: one key far above the rest, placed in a late tile (guides rule 26)."""
    H, L = 2, 640
    g = torch.Generator().manual_seed(5)
    q = torch.randn(H, L, 64, generator=g) * 0.125 * LOG2E
    k = torch.randn(H, L, 64, generator=g)
    v = torch.randn(H, L, 64, generator=g)
    k[:, 500] = q[:, 17] * 8 * 40.0  # row 17's score with key 500 is huge
    q, k, v = [_t16(x, BF16).to(dev) for x in (q, k, v)]
    o = torch.empty(L, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_attention(BF16, _p(q), _p(k), _p(v), _p(o), H, L, L, 1, 0, s) == 0
    torch.cuda.synchronize()
    got = _from16(o, BF16).reshape(L, H, 64)
    ref = _attn_ref(q.float(), k.float(), v.float()).transpose(0, 1)
    assert torch.isfinite(got).all()
    assert _rel(got, ref) < 8e-3


def test_attention_all_scores_far_below_zero(dev):
    """Every score is << 0 from the first tile on: the running max must still be tracked (first-tile forced
    update), otherwise every P underflows and O = 0/0."""
    H, L = 2, 200
    g = torch.Generator().manual_seed(11)
    base = torch.randn(H, 1, 64, generator=g)
    q = (base + 0.05 * torch.randn(H, L, 64, generator=g)) * 3.0
    k = -(base + 0.05 * torch.randn(H, L, 64, generator=g)) * 3.0   # q.k ~ -9 * |base|^2 ~ -600 (log2 units)
    v = torch.randn(H, L, 64, generator=g)
    q, k, v = [_t16(x, BF16).to(dev) for x in (q, k, v)]
    o = torch.empty(L, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_attention(BF16, _p(q), _p(k), _p(v), _p(o), H, L, L, 1, 0, s) == 0
    torch.cuda.synchronize()
    got = _from16(o, BF16).reshape(L, H, 64)
    ref = _attn_ref(q.float(), k.float(), v.float()).transpose(0, 1)
    assert torch.isfinite(got).all()
    assert _rel(got, ref) < 1e-2


def test_attention_kv_chunks_equals_concat(dev):
    """Sharded global attention: keys/values given as 2 gathered chunks == one concatenated sequence."""
    H, Lq, Lc = 4, 300, 300
    g = torch.Generator().manual_seed(9)
    q = _t16(torch.randn(H, Lq, 64, generator=g) * 0.125 * LOG2E, BF16).to(dev)
    kc = _t16(torch.randn(2, H, Lc, 64, generator=g), BF16).to(dev)
    vc = _t16(torch.randn(2, H, Lc, 64, generator=g), BF16).to(dev)
    o = torch.empty(Lq, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_attention(BF16, _p(q), _p(kc), _p(vc), _p(o), H, Lq, Lq, 2, Lc, s) == 0
    torch.cuda.synchronize()
    kk = torch.cat([kc[0], kc[1]], 1).float()
    vv = torch.cat([vc[0], vc[1]], 1).float()
    ref = _attn_ref(q.float(), kk, vv).transpose(0, 1)
    assert _rel(_from16(o, BF16).reshape(Lq, H, 64), ref) < 8e-3


@pytest.mark.parametrize("variant", [6, 10, 4])
def test_attention_variants(dev, variant):
    """The non-default kernels kept for A/B (6: software-pipelined half-tile kernel with LDS-DMA, 10: eager row max,
    4: 32 rows per wave) must stay correct: multi-tile, ragged tail, 2 sequences, plus the spike case in a late tile."""
    L_ = _lib()
    H, nseq, L = 4, 2, 1100
    M = nseq * L
    g = torch.Generator().manual_seed(variant)
    q = torch.randn(H, M, 64, generator=g) * 0.125 * 1.5 * LOG2E
    k = torch.randn(H, M, 64, generator=g) * 1.5
    v = torch.randn(H, M, 64, generator=g)
    k[:, L + 1000] = q[:, L + 17] * 8 * 30.0
    q, k, v = [_t16(x, BF16).to(dev) for x in (q, k, v)]
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L_.wm_set_tuning(b"attn_qb", variant) == 0
    try:
        assert L_.wm_op_attention(BF16, _p(q), _p(k), _p(v), _p(o), H, M, L, 1, 0, s) == 0
        torch.cuda.synchronize()
    finally:
        L_.wm_set_tuning(b"attn_qb", -1)
    got = _from16(o, BF16).reshape(M, H, 64)
    ref = torch.empty(M, H, 64, device=dev)
    for i in range(nseq):
        sl = slice(i * L, (i + 1) * L)
        ref[sl] = _attn_ref(q[:, sl].float(), k[:, sl].float(), v[:, sl].float()).transpose(0, 1)
    assert torch.isfinite(got).all()
    e = _rel(got, ref)
    print(f"attention variant {variant}: {e:.2e}")
    assert e < 8e-3


@pytest.mark.parametrize("H,L,chunks,splits", [(4, 2752, 1, 2), (2, 2200, 1, 4), (3, 1100, 2, 2), (2, 5504, 1, 3)])
def test_attention_split_kv(dev, H, L, chunks, splits):
    """Split-KV (partials + combine pass) == the single-pass kernel == fp32 softmax; the spike key sits in the LAST
    slice so the slices' running maxima differ by a lot (the combine must rescale, not just add)."""
    g = torch.Generator().manual_seed(H * 31 + L + splits)
    Lk = L // chunks if chunks > 1 else L
    q = torch.randn(H, L, 64, generator=g) * 0.125 * LOG2E
    k = torch.randn(chunks, H, Lk, 64, generator=g)
    v = torch.randn(chunks, H, Lk, 64, generator=g)
    k[-1, :, Lk - 5] = q[:, 33] * 8 * 30.0
    q, k, v = [_t16(x, BF16).to(dev) for x in (q, k, v)]
    o1 = torch.empty(L, H * 64, device=dev, dtype=torch.int16)
    o2 = torch.zeros(L, H * 64, device=dev, dtype=torch.int16)
    po = torch.full((splits, L, H * 64), float("nan"), device=dev)
    pml = torch.full((splits, H, L, 2), float("nan"), device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_ = _lib()
    assert L_.wm_op_attention(BF16, _p(q), _p(k), _p(v), _p(o1), H, L, L, chunks, Lk if chunks > 1 else 0, s) == 0
    assert L_.wm_op_attention_split(BF16, _p(q), _p(k), _p(v), _p(o2), H, L, L, chunks, Lk if chunks > 1 else 0, splits, _p(po), _p(pml), s) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(po).all() and torch.isfinite(pml).all(), "every slice must have written its partial"
    kk = torch.cat(list(k), 1).float()
    vv = torch.cat(list(v), 1).float()
    ref = _attn_ref(q.float(), kk, vv).transpose(0, 1)
    a1, a2 = _from16(o1, BF16).reshape(L, H, 64), _from16(o2, BF16).reshape(L, H, 64)
    print(f"split-kv H{H} L{L} chunks{chunks} splits{splits}: single {_rel(a1, ref):.2e} split {_rel(a2, ref):.2e} split-vs-single {_rel(a2, a1):.2e}")
    assert _rel(a2, ref) < 8e-3
    assert _rel(a2, a1) < 6e-3  # both round P and O to bf16, in different groupings


@pytest.mark.parametrize("H,nseq,L", [(16, 1, 11008), (16, 2, 4200), (40, 1, 4100), (33, 2, 5000)])
def test_attention_tail_split(dev, H, nseq, L):
    """Automatic tail split (one GPU, workspace given, kv_splits = 0): only the units of the launch's last, partly filled
    round are cut into key slices + combine pass; every other unit is written by its single block.  Checked against the
    unsplit kernel on the same inputs (every row, so whole and split units alike) and against fp32 softmax."""
    g = torch.Generator().manual_seed(H + nseq + L)
    R = nseq * L
    q = torch.randn(H, R, 64, generator=g) * 0.125 * LOG2E
    k = torch.randn(H, R, 64, generator=g)
    v = torch.randn(H, R, 64, generator=g)
    k[:, L - 3] = q[:, 40] * 8 * 30.0  # a spike key in the last slice of sequence 0
    q, k, v = [_t16(x, BF16).to(dev) for x in (q, k, v)]
    o1 = torch.zeros(R, H * 64, device=dev, dtype=torch.int16)
    o2 = torch.zeros(R, H * 64, device=dev, dtype=torch.int16)
    po = torch.zeros((8, R, H * 64), device=dev)
    pml = torch.zeros((8, H, R, 2), device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_ = _lib()
    assert L_.wm_op_attention(BF16, _p(q), _p(k), _p(v), _p(o1), H, R, L, 1, 0, s) == 0
    assert L_.wm_op_attention_split(BF16, _p(q), _p(k), _p(v), _p(o2), H, R, L, 1, 0, 0, _p(po), _p(pml), s) == 0
    torch.cuda.synchronize()
    units = H * nseq * ((L + 255) // 256)
    assert units > 512, "the case must span more than one round of 512 resident blocks"
    assert bool((po != 0).any()), "the tail units must have gone through the partial buffers"
    a1, a2 = _from16(o1, BF16).reshape(R, H, 64), _from16(o2, BF16).reshape(R, H, 64)
    same = (o1 == o2).reshape(R, H, 64).all(-1)  # [row][head]: whole units are bit-identical to the unsplit launch
    print(f"tail split H{H} nseq{nseq} L{L}: identical (row, head) pairs {float(same.float().mean()):.3f}, tail-vs-single {_rel(a2, a1):.2e}")
    assert 0.3 < float(same.float().mean()) < 1.0
    assert _rel(a2, a1) < 6e-3
    for hh in (0, H - 1):  # fp32 reference on the first (whole) and last (split) head, sequence 0
        ref = _attn_ref(q[hh:hh + 1, :L].float(), k[hh:hh + 1, :L].float(), v[hh:hh + 1, :L].float())[0]
        assert _rel(a2[:L, hh], ref) < 8e-3


def _attn_emulated(q, k, v, dt):
    """What the kernel computes, up to fp32 summation order (oracle/worldmirror_ref.py attention, emulation mode): base-2 softmax
    against an INTEGER row max, P rounded to 16 bits for P@V, fp32 row sums of the un-rounded P, O rounded to 16 bits."""
    S = q @ k.transpose(-1, -2)
    P = torch.exp2(S - torch.ceil(S.max(-1, keepdim=True)[0]))
    O = (_t16(P, dt).float() @ v) / P.sum(-1, keepdim=True)
    return _t16(O, dt).float()


@pytest.mark.parametrize("H,L,dt", [(16, 1376, BF16), (4, 2752, BF16), (4, 2752, F16)])
def test_attention_matches_emulated_rounding(dev, H, L, dt):
    """Kernel error proper: against the rounding-emulated softmax on the same operands the kernel's 16-bit output is
    IDENTICAL except where an fp32-level difference (summation order, v_exp_f32 vs exp2) flips a final rounding."""
    g = torch.Generator().manual_seed(H + L)
    q = _t16(torch.randn(H, L, 64, generator=g) * 0.125 * 1.5 * LOG2E, dt).to(dev)
    k = _t16(torch.randn(H, L, 64, generator=g) * 1.5, dt).to(dev)
    v = _t16(torch.randn(H, L, 64, generator=g), dt).to(dev)
    o = torch.empty(L, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_attention(dt, _p(q), _p(k), _p(v), _p(o), H, L, L, 1, 0, s) == 0
    torch.cuda.synchronize()
    got = _from16(o, dt).reshape(L, H, 64)
    emu = _attn_emulated(q.float(), k.float(), v.float(), dt).transpose(0, 1)
    diff = float((got != emu).float().mean())
    e = _rel(got, emu)
    print(f"attention vs emulated rounding H{H} L{L} dt{dt}: {100 * diff:.3f} % of the outputs differ (by one 16-bit ulp), rel-L2 {e:.2e}")
    assert diff < 0.02 and e < (3e-4 if dt == BF16 else 5e-5)


@pytest.mark.parametrize("H,L,chunks,splits", [(4, 2752, 1, 4), (3, 4400, 2, 3), (2, 5504, 4, 2)])
def test_attention_result_independent_of_key_partitioning(dev, H, L, chunks, splits):
    """The running max is an integer, so the 16-bit rounding of every softmax numerator is the same whatever the order,
    chunking (gathered shards) or KV split the keys are visited in: the outputs of the single-pass kernel, the chunked
    launch and the split-KV launch differ only where fp32 summation order flips a final 16-bit rounding."""
    g = torch.Generator().manual_seed(H * 7 + L)
    Lk = L // chunks
    q = _t16(torch.randn(H, L, 64, generator=g) * 0.125 * 1.5 * LOG2E, BF16).to(dev)
    k = _t16(torch.randn(chunks, H, Lk, 64, generator=g) * 1.5, BF16).to(dev)
    v = _t16(torch.randn(chunks, H, Lk, 64, generator=g), BF16).to(dev)
    kk = torch.cat(list(k), 1).contiguous()
    vv = torch.cat(list(v), 1).contiguous()
    outs = [torch.empty(L, H * 64, device=dev, dtype=torch.int16) for _ in range(3)]
    po = torch.zeros((splits, L, H * 64), device=dev)
    pml = torch.zeros((splits, H, L, 2), device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_ = _lib()
    assert L_.wm_op_attention(BF16, _p(q), _p(kk), _p(vv), _p(outs[0]), H, L, L, 1, 0, s) == 0
    assert L_.wm_op_attention(BF16, _p(q), _p(k), _p(v), _p(outs[1]), H, L, L, chunks, Lk if chunks > 1 else 0, s) == 0
    assert L_.wm_op_attention_split(BF16, _p(q), _p(k), _p(v), _p(outs[2]), H, L, L, chunks, Lk if chunks > 1 else 0, splits, _p(po), _p(pml), s) == 0
    torch.cuda.synchronize()
    a = [_from16(o, BF16) for o in outs]
    for nm, x in (("chunked", a[1]), ("split-KV", a[2])):
        diff, e = float((x != a[0]).float().mean()), _rel(x, a[0])
        print(f"{nm} vs single pass: {100 * diff:.3f} % of the outputs differ, rel-L2 {e:.2e}")
        assert diff < 0.02 and e < 3e-4


def _run_attn_ex(L_, q, k, v, H, R, Ls, chunks, Lc, splits, qb, dev, dt=BF16):
    o = torch.zeros(R, H * 64, device=dev, dtype=torch.int16)
    po = torch.zeros((8, R, H * 64), device=dev)
    pml = torch.zeros((8, H, R, 2), device=dev)
    flags = torch.full((int(L_.wm_op_attention_flag_count(R, Ls, H)),), -1, device=dev, dtype=torch.int32)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L_.wm_set_tuning(b"attn_qb", qb) == 0
    try:
        assert L_.wm_op_attention_ex(dt, _p(q), _p(k), _p(v), _p(o), H, R, Ls, chunks, Lc, splits, _p(po), _p(pml), _p(flags), s) == 0
        torch.cuda.synchronize()
    finally:
        L_.wm_set_tuning(b"attn_qb", -1)
    return _from16(o, dt).reshape(R, H, 64), flags[: (flags.numel() - 4) // 2]   # ([flags | sticky hints | counter])


@pytest.mark.parametrize("H,nseq,L,chunks,splits", [(4, 1, 2752, 1, 1), (16, 1, 11008, 1, 0), (3, 2, 1408, 1, 1), (4, 1, 2816, 2, 2), (2, 1, 4096, 4, 0),
                                                     (16, 8, 1376, 1, 0), (16, 3, 1374, 1, 0), (2, 1, 1000, 1, 1), (3, 2, 577, 1, 1),
                                                     (3, 1, 4128, 3, 0), (2, 1, 9632, 1, 0)])
def test_attention_v3_pipelined_no_max_kernel(dev, H, nseq, L, chunks, splits):
    """attention_v3.hip (attn_qb = 7): software-pipelined, no running max.  Same operands -> the general kernel's result up to
    final-rounding flips (2^S / sum 2^S is scale-free, bf16 rounding of P too), fp32 softmax within the bf16 P / O rounding;
    whole units, uniform splits (chunks > 1), the tail split (16 heads x 43 q-tiles = 688 units on 512 slots), 2 sequences, and
    RAGGED sequences — per-frame 1376 = 21.5 key tiles (8 frames x 6 q-tiles x 16 heads = 768 units: tail split in halves),
    DINO's 1374 (the last tile has 30 keys), 1000 and 577 rows, three gathered chunks of one view each (every chunk ends in a
    ragged tile) and a 7-view cross-view sequence (150.5 tiles).  Ragged tiles are padded with zero keys by the DMA (P = 1 exactly,
    V = 0) and the pads are taken out of the row sums in the epilogue: round 3, no masked instantiation any more."""
    L_ = _lib()
    g = torch.Generator().manual_seed(H * 13 + L + chunks)
    R = nseq * L
    Lc = L // chunks if chunks > 1 else 0
    q = _t16(torch.randn(H, R, 64, generator=g) * 0.125 * 1.5 * LOG2E, BF16).to(dev)
    if chunks > 1:
        k = _t16(torch.randn(chunks, H, Lc, 64, generator=g) * 1.5, BF16).to(dev)
        v = _t16(torch.randn(chunks, H, Lc, 64, generator=g), BF16).to(dev)
    else:
        k = _t16(torch.randn(H, R, 64, generator=g) * 1.5, BF16).to(dev)
        v = _t16(torch.randn(H, R, 64, generator=g), BF16).to(dev)
    a3, _ = _run_attn_ex(L_, q, k, v, H, R, L, chunks, Lc, splits, 3, dev)
    a7, flags = _run_attn_ex(L_, q, k, v, H, R, L, chunks, Lc, splits, 7, dev)
    used = flags[flags >= 0]
    assert used.numel() > 0, "the v3 kernel did not run (launcher fell back)"
    assert int(used.sum()) == 0, "no block may leave the no-max range on bounded scores"
    diff, e = float((a7 != a3).float().mean()), _rel(a7, a3)
    print(f"v3 vs general H{H} nseq{nseq} L{L} chunks{chunks} splits{splits}: {100 * diff:.3f} % of the outputs differ, rel-L2 {e:.2e}; blocks {used.numel()}")
    assert torch.isfinite(a7).all() and diff < 0.02 and e < 3e-4
    kk = (torch.cat(list(k), 1) if chunks > 1 else k).float()
    vv = (torch.cat(list(v), 1) if chunks > 1 else v).float()
    for i in range(nseq):
        sl = slice(i * L, (i + 1) * L)
        ks = kk if chunks > 1 else kk[:, sl]
        vs = vv if chunks > 1 else vv[:, sl]
        ref = _attn_ref(q[:, sl].float(), ks, vs).transpose(0, 1)
        assert _rel(a7[sl], ref) < 8e-3


@pytest.mark.parametrize("kind", ["spike_overflow", "all_far_below_zero"])
def test_attention_v3_out_of_range_rows_are_recomputed(dev, kind):
    """The no-max form is only valid while every row sum stays a comfortably normal fp32 number.  Rows that leave
    [2^-80, 2^100] — a score of +200 (2^200 overflows), or every score near -600 (every 2^S flushes to 0) — must raise their
    block's flag and come out right from the general kernel's recompute pass; all other blocks are left to the fast kernel."""
    L_ = _lib()
    H, L = 2, 2048
    g = torch.Generator().manual_seed(3)
    q = torch.randn(H, L, 64, generator=g) * 0.125 * LOG2E
    k = torch.randn(H, L, 64, generator=g)
    v = torch.randn(H, L, 64, generator=g)
    if kind == "spike_overflow":
        k[0, 1500] = q[0, 700] * 8 * 60.0        # head 0, query row 700 (q-tile 2): one score ~ +300 in log2 units
        expect = {(0, 2)}
    else:
        base = torch.randn(64, generator=g)
        q[1, 256:512] = (base + 0.05 * torch.randn(256, 64, generator=g)) * 3.0   # head 1, q-tile 1: every score ~ -600
        k[1] = -(base + 0.05 * torch.randn(L, 64, generator=g)) * 3.0
        expect = {(1, 1)}
    q, k, v = [_t16(x, BF16).to(dev) for x in (q, k, v)]
    a7, flags = _run_attn_ex(L_, q, k, v, H, L, L, 1, 0, 1, 7, dev)
    ref = _attn_ref(q.float(), k.float(), v.float()).transpose(0, 1)
    assert torch.isfinite(a7).all()
    assert _rel(a7, ref) < 1e-2
    used = flags[: H * (L // 256)]
    assert int(used.sum()) >= 1 and int(used.sum()) <= 2 * len(expect) + 6, used.tolist()   # only the affected units (+ their head's neighbours at most)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("H,nseq,L,chunks,splits", [(4, 1, 2752, 1, 1), (16, 1, 11008, 1, 0), (3, 2, 1408, 1, 1), (4, 1, 2816, 2, 2), (2, 1, 4096, 4, 0),
                                                     (2, 1, 576, 1, 1), (5, 3, 1024, 1, 0),
                                                     (16, 8, 1376, 1, 0), (16, 3, 1374, 1, 0), (2, 1, 1000, 1, 1), (3, 1, 4128, 3, 0), (2, 1, 9632, 1, 0)])
def test_attention_v4_one_wave_per_simd(dev, H, nseq, L, chunks, splits, dt):
    """attention_v4.hip (attn_qb = 8): one wave per SIMD, 128 query rows per wave, every MFMA gap a hand-placed asm statement; bf16
    without a running max, f16 with the row max fixed from the first key tile (+ headroom) in the QK chain's initial accumulator.
    Same operands -> the general kernel's result up to final-rounding flips (integer max: the mantissa of P does not depend on it),
    fp32 softmax within the 16-bit P / O rounding.  Whole units, uniform splits, chunked keys, the tail split (16 heads x 22 q-tiles
    = 352 units on 256 slots), several sequences, a 576-key sequence (9 tiles: the shortest pipelines), a last q-tile with 64 of
    512 rows (576 = 512 + 64); ragged segments (per-frame 1376, DINO 1374, 1000 rows, three one-view chunks, 7 views): the last
    tile of a segment is padded with zero keys (P = 2^0 / 2^-m exactly, V = 0), taken out of the row sums in the epilogue."""
    L_ = _lib()
    g = torch.Generator().manual_seed(H * 13 + L + chunks)
    R = nseq * L
    Lc = L // chunks if chunks > 1 else 0
    q = _t16(torch.randn(H, R, 64, generator=g) * 0.125 * 1.5 * LOG2E, dt).to(dev)
    if chunks > 1:
        k = _t16(torch.randn(chunks, H, Lc, 64, generator=g) * 1.5, dt).to(dev)
        v = _t16(torch.randn(chunks, H, Lc, 64, generator=g), dt).to(dev)
    else:
        k = _t16(torch.randn(H, R, 64, generator=g) * 1.5, dt).to(dev)
        v = _t16(torch.randn(H, R, 64, generator=g), dt).to(dev)
    a3, _ = _run_attn_ex(L_, q, k, v, H, R, L, chunks, Lc, splits, 3, dev, dt)
    a8, flags = _run_attn_ex(L_, q, k, v, H, R, L, chunks, Lc, splits, 8, dev, dt)
    used = flags[flags >= 0]
    assert used.numel() > 0, "the v4 kernel did not run (launcher fell back)"
    assert int(used.sum()) == 0, "no unit may be flagged on bounded scores"
    diff, e = float((a8 != a3).float().mean()), _rel(a8, a3)
    print(f"v4 vs general dt{dt} H{H} nseq{nseq} L{L} chunks{chunks} splits{splits}: {100 * diff:.3f} % of the outputs differ, rel-L2 {e:.2e}; blocks {used.numel()}")
    assert torch.isfinite(a8).all() and diff < (0.02 if dt == BF16 else 0.06) and e < (3e-4 if dt == BF16 else 8e-5)
    kk = (torch.cat(list(k), 1) if chunks > 1 else k).float()
    vv = (torch.cat(list(v), 1) if chunks > 1 else v).float()
    for i in range(nseq):
        sl = slice(i * L, (i + 1) * L)
        ks = kk if chunks > 1 else kk[:, sl]
        vs = vv if chunks > 1 else vv[:, sl]
        ref = _attn_ref(q[:, sl].float(), ks, vs).transpose(0, 1)
        assert _rel(a8[sl], ref) < (8e-3 if dt == BF16 else 1e-3)


@pytest.mark.parametrize("dt", [BF16, F16])
def test_attention_v4_matches_emulated_rounding(dev, dt):
    """Kernel error proper for attn_v4: against the rounding-emulated softmax (integer row max, P rounded to 16 bits, fp32 row
    sums) on the same operands the 16-bit output is identical except for final-rounding flips.  f16: the emulation subtracts the
    TRUE integer row max, the kernel a max fixed after 64 keys + 4: a different power of two, i.e. the same mantissas — only
    P below 2^-14 of the respective reference point round differently (subnormal), far below the output's ulp."""
    L_ = _lib()
    H, L = 4, 2752
    g = torch.Generator().manual_seed(H + L)
    q = _t16(torch.randn(H, L, 64, generator=g) * 0.125 * 1.5 * LOG2E, dt).to(dev)
    k = _t16(torch.randn(H, L, 64, generator=g) * 1.5, dt).to(dev)
    v = _t16(torch.randn(H, L, 64, generator=g), dt).to(dev)
    got, flags = _run_attn_ex(L_, q, k, v, H, L, L, 1, 0, 1, 8, dev, dt)
    assert int(flags[flags >= 0].sum()) == 0
    emu = _attn_emulated(q.float(), k.float(), v.float(), dt).transpose(0, 1)
    diff, e = float((got != emu).float().mean()), _rel(got, emu)
    print(f"v4 vs emulated rounding dt{dt}: {100 * diff:.3f} % of the outputs differ (by one 16-bit ulp), rel-L2 {e:.2e}")
    assert diff < (0.02 if dt == BF16 else 0.06) and e < (3e-4 if dt == BF16 else 8e-5)


@pytest.mark.parametrize("dt,kind", [(BF16, "spike_overflow"), (BF16, "all_far_below_zero"), (F16, "late_spike"), (F16, "all_far_below_zero")])
def test_attention_v4_out_of_range_rows_are_recomputed(dev, dt, kind):
    """The fast forms are only valid inside a range that is checked, not assumed: bf16 — every row sum in [2^-80, 2^100]; f16 — no
    P beyond f16 (a score more than 2^20 above what the row's first 64 keys showed: O turns non-finite) and a row sum that is a
    normal number.  A unit with such a row raises its flag and comes out right from the general kernel's recompute pass; all
    other units are left to the fast kernel."""
    L_ = _lib()
    H, L = 2, 2048
    g = torch.Generator().manual_seed(3)
    q = torch.randn(H, L, 64, generator=g) * 0.125 * LOG2E
    k = torch.randn(H, L, 64, generator=g)
    v = torch.randn(H, L, 64, generator=g)
    if kind == "spike_overflow":
        k[0, 1500] = q[0, 700] * 8 * 60.0        # head 0, query row 700 (unit 1): one score ~ +300 in log2 units
        expect = {(0, 1)}
    elif kind == "late_spike":
        k[0, 1500] = q[0, 700] * 8 * 9.0         # ~ +45 log2 units at key 1500: far above the window set after the first 64 keys
        expect = {(0, 1)}
    else:
        base = torch.randn(64, generator=g)
        q[1, 512:1024] = (base + 0.05 * torch.randn(512, 64, generator=g)) * 3.0   # head 1, unit 1: every score ~ -600
        k[1] = -(base + 0.05 * torch.randn(L, 64, generator=g)) * 3.0
        expect = {(1, 1)}
    q, k, v = [_t16(x, dt).to(dev) for x in (q, k, v)]
    a8, flags = _run_attn_ex(L_, q, k, v, H, L, L, 1, 0, 1, 8, dev, dt)
    ref = _attn_ref(q.float(), k.float(), v.float()).transpose(0, 1)
    assert torch.isfinite(a8).all()
    assert _rel(a8, ref) < (1e-2 if dt == BF16 else 2e-3)
    used = flags[: H * (L // 512)]
    nflag = int((used != 0).sum())
    print(f"v4 {kind} dt{dt}: flags {used.tolist()}")
    if not (dt == F16 and kind == "all_far_below_zero"):   # (f16 subtracts a max: uniformly low scores are in range there)
        assert 1 <= nflag <= 2 * len(expect) + 2, used.tolist()


@pytest.mark.parametrize("qb", [8, 3])
@pytest.mark.parametrize("kind", ["sink_in_first_tile_bulk_18_below", "wide_gaussian_sigma_7", "uniformly_low_ragged_1376"])
def test_attention_f16_wide_score_spread(dev, qb, kind):
    """f16 P = 2^(S - m) has 5 exponent bits: with m fixed from the first key tile (attn_v4) the keys far below it lose bits or flush in
    O while the fp32 row sum still counts them.  Three distributions with a NON-ZERO mean in V (so lost mass shows):
    a sink in the first tile 18 log2 units above 22 K bulk keys (8 % of the row's mass sits in the bulk), gaussian scores with a
    sigma of 7 log2 units, and per-frame 1376-key rows (32 zero pads in the ragged last tile) whose scores all sit 10 - 14 log2
    units below zero.  fp32 softmax at the f16 tolerance; the general kernel (3) is held to the same bound."""
    L_ = _lib()
    H = 2
    g = torch.Generator().manual_seed(29)
    u = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
    if kind == "sink_in_first_tile_bulk_18_below":
        L, nseq = 16 * 1376, 1
        q = torch.randn(H, L, 64, generator=g) * 0.125 * 1.2 * LOG2E + 1.0 * u
        k = torch.randn(H, L, 64, generator=g) * 1.2
        k[:, 3] = u * 18.0
    elif kind == "wide_gaussian_sigma_7":
        L, nseq = 8 * 1376, 1
        q = torch.randn(H, L, 64, generator=g) * 0.5
        k = torch.randn(H, L, 64, generator=g) * 1.75     # q.k ~ N(0, 64 * 0.5^2 * 1.75^2) = N(0, 7^2) in log2 units
    else:
        L, nseq = 1376, 4
        q = torch.randn(H, L * nseq, 64, generator=g) * 0.02 + 1.0 * u
        k = torch.randn(H, L * nseq, 64, generator=g) * 0.5 - 12.0 * u
    R = L * nseq
    v = torch.randn(H, R, 64, generator=g) + 1.0
    q, k, v = [_t16(x, F16).to(dev) for x in (q, k, v)]
    got, flags = _run_attn_ex(L_, q, k, v, H, R, L, 1, 0, 1, qb, dev, F16)
    errs = []
    for i in range(nseq):
        sl = slice(i * L, (i + 1) * L)
        ref = _attn_ref(q[:, sl].float(), k[:, sl].float(), v[:, sl].float()).transpose(0, 1)
        errs.append(_rel(got[sl], ref))
    nflag = int((flags[flags >= 0] != 0).sum()) if qb == 8 else 0
    print(f"f16 {kind} qb{qb}: rel-L2 {max(errs):.2e}, flagged units {nflag}")
    assert torch.isfinite(got).all() and max(errs) < 2e-3


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("where", ["first_tile", "late"])
def test_attention_sink_shaped_scores_and_sticky_hint(dev, dt, where):
    """Attention-sink logits (a few keys far above the rest for EVERY query, as register / camera tokens of a trained ViT produce):
    in the first key tile the fast kernels must take them in their stride (bf16: 2^S stays far inside fp32; f16: the fixed row max
    comes from that tile); arriving late and beyond the fast range (bf16: 2^130 overflows the row sum; f16: 2^50 above the window
    set after 64 keys) every unit is flagged and recomputed — and REMEMBERED: on the following calls the unit's hint sends it to
    the general kernel alone (flag 17 = by hint), counting down WM_ATTN_HINT_TTL until the fast kernel is tried again.  Results
    against the fp32 softmax on every call."""
    L_ = _lib()
    H, L = 4, 4096
    g = torch.Generator().manual_seed(11 + dt)
    u = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
    q = torch.randn(H, L, 64, generator=g) * 0.125 * 1.5 * LOG2E + 2.0 * u
    k = torch.randn(H, L, 64, generator=g) * 1.5
    v = torch.randn(H, L, 64, generator=g)
    gain = (65.0 if dt == BF16 else 25.0) if where == "late" else (30.0 if dt == BF16 else 20.0)   # score = gain * (2 + noise) log2 units
    pos = [3000, 3001, 3002, 4090] if where == "late" else [0, 5, 17, 40]
    for h in range(H):
        for j, ps in enumerate(pos): k[h, ps] = u * gain * (1.0 - 0.03 * j)
    q, k, v = [_t16(x, dt).to(dev) for x in (q, k, v)]
    ref = _attn_ref(q.float(), k.float(), v.float()).transpose(0, 1)
    o = torch.zeros(L, H * 64, device=dev, dtype=torch.int16)
    po = torch.zeros((8, L, H * 64), device=dev); pml = torch.zeros((8, H, L, 2), device=dev)
    ntot = int(L_.wm_op_attention_flag_count(L, L, H))
    n = ntot - 4                                               # [flags | hints | counter + pad]
    buf = torch.zeros((ntot,), device=dev, dtype=torch.int32)  # kept across the calls like the forward's workspace
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L_.wm_set_tuning(b"attn_qb", 8) == 0
    try:
        seen = []
        for call in range(4):
            buf[: n // 2] = -1
            o.zero_()
            assert L_.wm_op_attention_ex(dt, _p(q), _p(k), _p(v), _p(o), H, L, L, 1, 0, 1, _p(po), _p(pml), _p(buf), s) == 0
            torch.cuda.synchronize()
            got = _from16(o, dt).reshape(L, H, 64)
            e = _rel(got, ref)
            flags = buf[: n // 2]; used = flags[flags >= 0]; hints = buf[n // 2: n]
            seen.append((sorted(set(used.tolist())), int(hints.max())))
            print(f"sink {where} dt{dt} call {call}: rel-L2 {e:.2e}, flag values {seen[-1][0]}, max hint {seen[-1][1]}")
            assert torch.isfinite(got).all() and e < (1e-2 if dt == BF16 else 2e-3)
            assert used.numel() == H * (L // 512)
        if where == "first_tile":
            assert all(f == [0] and hmax == 0 for f, hmax in seen), "sinks inside the first key tile must stay on the fast kernel"
        else:
            assert 0 not in seen[0][0] and 17 not in seen[0][0] and seen[0][1] == 15      # every unit flagged by the computation, remembered
            for call in (1, 2, 3):
                assert seen[call][0] == [17] and seen[call][1] == 15 - call               # by hint only, counting down
    finally:
        L_.wm_set_tuning(b"attn_qb", -1)


@pytest.mark.parametrize("D", [128, 256, 1024, 2048])
def test_layernorm(dev, D):
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(333, D, generator=g) * 3 + 1).to(dev)
    w = torch.randn(D, generator=g).to(dev)
    b = torch.randn(D, generator=g).to(dev)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty_like(x)
    assert _lib().wm_op_layernorm(_p(x), _p(out), _p(w), _p(b), 333, D, 1e-5, 1, 0, s) == 0
    o16 = torch.empty(333, D, device=dev, dtype=torch.int16)
    assert _lib().wm_op_layernorm(_p(x), _p(o16), _p(w), _p(b), 333, D, 1e-5, 0, BF16, s) == 0
    torch.cuda.synchronize()
    assert _rel(out, ref) < 2e-6
    assert _rel(_from16(o16, BF16), ref) < 4e-3
    # D = 1024 / 2048 with a 16-bit output take the branch-free kernel (every load up front): same expressions, same bits
    assert torch.equal(_from16(o16, BF16), out.to(torch.bfloat16).float())


def test_qkv_post_matches_oracle(dev):
    """per-head LayerNorm(64) + 2-D RoPE + head-major relayout vs the oracle's rope_2d/layer_norm."""
    from oracle import worldmirror_ref as R
    H, S, gh, gw, psi = 4, 2, 4, 5, 7
    P = psi + gh * gw
    M, D = S * P, H * 64
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(M, 3 * D, generator=g)
    nw = [torch.randn(64, generator=g) for _ in range(4)]
    cos_t, sin_t = R.rope_tables(max(gh, gw) + 1, 32, 100.0)
    cos16, sin16 = cos_t[:, :16].contiguous(), sin_t[:, :16].contiguous()
    yy, xx = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.cat([torch.zeros(psi, 2, dtype=torch.long), torch.stack([yy.flatten(), xx.flatten()], -1) + 1], 0)
    pos = pos[None].expand(S, -1, -1).reshape(1, M, 2)
    t = qkv.reshape(1, M, 3, H, 64).permute(2, 0, 3, 1, 4)
    q = R.rope_2d(R.layer_norm(t[0], nw[0], nw[1], 1e-5), pos, 100.0) * 0.125
    k = R.rope_2d(R.layer_norm(t[1], nw[2], nw[3], 1e-5), pos, 100.0)
    v = t[2]
    d = [x.to(dev) for x in (qkv, *nw, cos16, sin16)]
    outs = [torch.empty(H, M, 64, device=dev, dtype=torch.int16) for _ in range(3)]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_qkv_post(F16, _p(d[0]), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]),
                                 _p(d[5]), _p(d[6]), M, H, P, psi, gw, 0.125, s) == 0
    torch.cuda.synchronize()
    for got, ref, nm in zip(outs, (q, k, v), "qkv"):
        e = _rel(_from16(got, F16).cpu(), ref[0])
        print("qkv_post", nm, e)
        assert e < 6e-4, nm


@pytest.mark.parametrize("H,S,gh,gw", [(4, 2, 4, 5), (16, 2, 37, 37), (2, 3, 5, 4)])
def test_gemm_qkv_fused_epilogue(dev, H, S, gh, gw):
    """QKV GEMM with fused per-head LayerNorm + 2-D RoPE + relayout == oracle ops on the fp32 GEMM result."""
    from oracle import worldmirror_ref as R
    psi = 7
    P = psi + gh * gw
    M, D, K = S * P, H * 64, 128
    g = torch.Generator().manual_seed(H + gh)
    A = _t16(torch.randn(M, K, generator=g), BF16)
    W = _t16(torch.randn(3 * D, K, generator=g) / math.sqrt(K), BF16)
    bias = torch.randn(3 * D, generator=g) * 0.1
    nw = [torch.randn(64, generator=g) for _ in range(4)]
    qkv = A.float() @ W.float().t() + bias
    cos_t, sin_t = R.rope_tables(max(gh, gw) + 1, 32, 100.0)
    cos16, sin16 = cos_t[:, :16].contiguous(), sin_t[:, :16].contiguous()
    yy, xx = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.cat([torch.zeros(psi, 2, dtype=torch.long), torch.stack([yy.flatten(), xx.flatten()], -1) + 1], 0)
    pos = pos[None].expand(S, -1, -1).reshape(1, M, 2)
    t = qkv.reshape(1, M, 3, H, 64).permute(2, 0, 3, 1, 4)
    q = R.rope_2d(R.layer_norm(t[0], nw[0], nw[1], 1e-5), pos, 100.0) * 0.25
    k = R.rope_2d(R.layer_norm(t[1], nw[2], nw[3], 1e-5), pos, 100.0)
    v = t[2]
    d = [x.to(dev) for x in (A, W, bias, *nw, cos16, sin16)]
    outs = [torch.empty(H, M, 64, device=dev, dtype=torch.int16) for _ in range(3)]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_gemm_qkv(BF16, _p(d[0]), _p(d[1]), _p(d[2]), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(d[3]), _p(d[4]),
                                 _p(d[5]), _p(d[6]), _p(d[7]), _p(d[8]), M, H, K, P, psi, gw, 0.25, s) == 0
    torch.cuda.synchronize()
    for got, ref, nm in zip(outs, (q, k, v), "qkv"):
        e = _rel(_from16(got, BF16).cpu(), ref[0])
        print("gemm_qkv", nm, f"{e:.2e}")
        assert e < 4e-3, nm   # one bf16 output rounding
    # without norm / rope (the DINO blocks)
    assert _lib().wm_op_gemm_qkv(BF16, _p(d[0]), _p(d[1]), _p(d[2]), _p(outs[0]), _p(outs[1]), _p(outs[2]), None, None, None, None,
                                 None, None, M, H, K, P, 0, gw, 0.25, s) == 0
    torch.cuda.synchronize()
    for got, ref in zip(outs, (t[0] * 0.25, t[1], t[2])):
        assert _rel(_from16(got, BF16).cpu(), ref[0]) < 4e-3


def _conv_ref(x, w, b, stride, pad, relu_in, resid, resid_relu, resid2, dt):
    xin = torch.relu(x) if relu_in else x
    xin = _t16(xin, dt).float()
    y = torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2), w, b, stride=stride, padding=pad).permute(0, 2, 3, 1)
    if resid is not None:
        y = y + (torch.relu(resid) if resid_relu else resid)
    if resid2 is not None:
        y = y + resid2
    return y


@pytest.mark.parametrize("Cin,Cout,ks,stride,Hh,Ww", [(64, 64, 3, 1, 20, 16), (256, 256, 3, 1, 37, 37), (32, 32, 3, 1, 70, 56),
                                                      (128, 32, 3, 1, 30, 30), (256, 256, 3, 2, 37, 37), (64, 64, 1, 1, 10, 8),
                                                      (512, 256, 3, 1, 10, 10), (256, 128, 3, 1, 40, 40),
                                                      (1024, 256, 3, 1, 16, 16), (512, 256, 3, 1, 32, 32), (256, 256, 3, 1, 64, 64),
                                                      (256, 128, 3, 1, 128, 128), (128, 32, 3, 1, 224, 224), (256, 256, 1, 1, 8, 8),
                                                      (256, 256, 1, 1, 64, 64), (1024, 1024, 3, 2, 16, 16)])
def test_conv(dev, Cin, Cout, ks, stride, Hh, Ww):
    dt = F16
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(Cin + Cout + ks)
    N = 2
    x = torch.randn(N, Hh, Ww, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks))
    w = _t16(w, dt).float().to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    Ho, Wo = (Hh + 2 * pad - ks) // stride + 1, (Ww + 2 * pad - ks) // stride + 1
    resid = torch.randn(N, Ho, Wo, Cout, generator=g).to(dev)
    resid2 = torch.randn(N, Ho, Wo, Cout, generator=g).to(dev)
    w16 = _t16(w.permute(0, 2, 3, 1).contiguous(), dt)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for relu_in, use_res in ((0, False), (1, True)):
        y = torch.empty(N, Ho, Wo, Cout, device=dev)
        st = _lib().wm_op_conv(dt, _p(x), _p(w16), _p(b), _p(resid) if use_res else None, _p(resid2) if use_res else None, _p(y),
                               N, Hh, Ww, Cin, Cout, ks, stride, pad, relu_in, 1 if use_res else 0, s)
        assert st == 0
        torch.cuda.synchronize()
        ref = _conv_ref(x, w, b, stride, pad, relu_in, resid if use_res else None, True, resid2 if use_res else None, dt)
        e = _rel(y, ref)
        print(f"conv {Cin}->{Cout} k{ks}s{stride} relu{relu_in}: {e:.2e}")
        assert e < 2e-5


@pytest.mark.parametrize("Cin,Cout,ks,Hh,Ww", [(256, 256, 3, 37, 37), (256, 128, 3, 40, 40), (64, 64, 3, 20, 16), (256, 256, 1, 16, 16), (128, 32, 3, 30, 30)])
@pytest.mark.parametrize("scale", [1e5, 1e-6])
def test_conv_f16_operand_range(dev, Cin, Cout, ks, Hh, Ww, scale):
    """The reference's DPT heads are fp32 (worldmirror.py:146): activations of a real checkpoint may leave f16's range.  The f16
    staging SATURATES at +-65504 (wm_common.h f2h): inputs of magnitude 1e5 give finite outputs equal to the conv of the clamped
    input (an unsaturated cast would feed inf to the MFMA: NaN rows); inputs of magnitude 1e-6 land in f16's subnormals and match
    the conv of the f16-rounded input.  With head_dtype = bf16 the same inputs are in range: error = plain bf16 operand rounding."""
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(Cin + Cout + ks)
    N = 2
    x = (torch.randn(N, Hh, Ww, Cin, generator=g) * scale).to(dev)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = (torch.randn(Cout, generator=g) * scale).to(dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dt in (F16, BF16):
        wq = _t16(w, dt).float().to(dev)
        w16 = _t16(wq.permute(0, 2, 3, 1).contiguous(), dt)
        y = torch.empty(N, Hh, Ww, Cout, device=dev)
        assert _lib().wm_op_conv(dt, _p(x), _p(w16), _p(b), None, None, _p(y), N, Hh, Ww, Cin, Cout, ks, 1, pad, 0, 0, s) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(y).all(), (dt, scale)
        xin = x.clamp(-65504.0, 65504.0) if dt == F16 else x
        ref = torch.nn.functional.conv2d(_t16(xin, dt).float().permute(0, 3, 1, 2), wq, b, padding=pad).permute(0, 2, 3, 1)
        e = _rel(y, ref)
        full = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), wq, b, padding=pad).permute(0, 2, 3, 1)
        print(f"conv {Cin}->{Cout} k{ks} x{scale:g} dt{dt}: vs same-rounding reference {e:.2e}, vs fp32-activation conv {_rel(y, full):.2e}")
        assert e < 2e-5
        if dt == BF16:
            assert _rel(y, full) < 4e-3  # in range: 8-bit mantissa rounding of the activations only


@pytest.mark.parametrize("Cin,Cout,ks,Hh,Ww", [(256, 256, 3, 37, 37), (128, 32, 3, 30, 30), (256, 256, 1, 16, 16)])
def test_conv_f16_staging_keeps_nan(dev, Cin, Cout, ks, Hh, Ww):
    """The saturating f16 staging must not hide an upstream fault: the reference's fp32 heads return NaN when an activation is NaN;
    v_med3 alone orders NaN low and would hand the MFMA -65504 — plausible finite garbage.  f2h keeps a NaN a NaN: every output
    pixel whose 3 x 3 (1 x 1) window holds the NaN input is NaN, every other one is finite."""
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(Cin + ks)
    N = 1
    x = torch.randn(N, Hh, Ww, Cin, generator=g)
    py, px, pc = Hh // 2, Ww // 3, 5
    x[0, py, px, pc] = float("nan")
    x = x.to(dev)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w16 = _t16(w.permute(0, 2, 3, 1).contiguous(), F16).to(dev)
    y = torch.empty(N, Hh, Ww, Cout, device=dev)
    assert _lib().wm_op_conv(F16, _p(x), _p(w16), None, None, None, _p(y), N, Hh, Ww, Cin, Cout, ks, 1, pad, 0, 0, s) == 0
    torch.cuda.synchronize()
    nan_px = torch.isnan(y).any(-1)[0]
    want = torch.zeros(Hh, Ww, dtype=torch.bool, device=dev)
    want[max(py - pad, 0):py + pad + 1, max(px - pad, 0):px + pad + 1] = True
    assert torch.equal(nan_px, want), f"NaN pixels {int(nan_px.sum())}, expected {int(want.sum())}"


@pytest.mark.parametrize("Cin,Cout,Hh,Ww", [(256, 256, 37, 37), (256, 256, 70, 45), (256, 128, 40, 40), (128, 128, 33, 100), (256, 256, 148, 148)])
def test_conv3x3_wide_pixel_tile(dev, Cin, Cout, Hh, Ww):
    """conv3x3_rs_kernel with 32 x 8 pixel tiles (chosen automatically where it saves a round over the CUs, e.g. 148^2
    at 8 views; forced here with conv_tpx = 32) against the fp32 torch reference and against the 16 x 16 tiling."""
    dt = F16
    g = torch.Generator().manual_seed(Cin + Cout + Hh)
    N = 2
    x = torch.randn(N, Hh, Ww, Cin, generator=g).to(dev)
    w = _t16(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dt).float().to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    resid = torch.randn(N, Hh, Ww, Cout, generator=g).to(dev)
    resid2 = torch.randn(N, Hh, Ww, Cout, generator=g).to(dev)
    w16 = _t16(w.permute(0, 2, 3, 1).contiguous(), dt)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = _conv_ref(x, w, b, 1, 1, 1, resid, True, resid2, dt)
    outs = {}
    try:
        for tpx in (32, 16):
            assert _lib().wm_set_tuning(b"conv_tpx", tpx) == 0
            y = torch.full((N, Hh, Ww, Cout), float("nan"), device=dev)
            assert _lib().wm_op_conv(dt, _p(x), _p(w16), _p(b), _p(resid), _p(resid2), _p(y), N, Hh, Ww, Cin, Cout, 3, 1, 1, 1, 1, s) == 0
            torch.cuda.synchronize()
            assert _rel(y, ref) < 2e-5, (tpx, _rel(y, ref))
            outs[tpx] = y
    finally:
        _lib().wm_set_tuning(b"conv_tpx", -1)
    assert torch.equal(outs[16], outs[32])  # same accumulation order per output element


@pytest.mark.parametrize("N,Hs,Ws,Hi,Wi,Cin", [(2, 20, 16, 35, 28, 64), (3, 40, 32, 70, 56, 128), (1, 9, 11, 33, 40, 128), (2, 74, 74, 130, 130, 128)])
def test_up_conv_n32_unfused(dev, N, Hs, Ws, Hi, Wi, Cin):
    """wm_op_up_conv_n32: align_corners resize (+ position tables) written as f16, then the 32-channel 3x3 conv fed by LDS-DMA
    (conv_n32.hip, persistent blocks, deferred stores) — against fp32 torch on the same f16-rounded operands and against the
    fused-resize kernel (wm_op_conv3x3_up); ragged tiles, more tiles than blocks (130^2: 81 tiles per image), bit-exact repeat."""
    g = torch.Generator().manual_seed(N + Hs + Cin)
    x = torch.randn(N, Hs, Ws, Cin, generator=g).to(dev)
    w = (torch.randn(32, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).half().to(dev)
    b = torch.randn(32, generator=g).to(dev)
    ax = torch.randn(Wi, Cin // 2, generator=g).to(dev); ay = torch.randn(Hi, Cin // 2, generator=g).to(dev)
    y1 = torch.empty(N, Hi, Wi, 32, device=dev); y2 = torch.full((N, Hi, Wi, 32), float("nan"), device=dev)
    up16 = torch.empty(N * Hi * Wi * Cin + 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_ = _lib()
    assert L_.wm_op_conv3x3_up(F16, _p(x), _p(w.view(torch.int16)), _p(b), _p(y1), N, Hs, Ws, Hi, Wi, Cin, 32, _p(ax), _p(ay), s) == 0
    for relu in (0, 1):
        assert L_.wm_op_up_conv_n32(F16, _p(x), _p(w.view(torch.int16)), _p(b), _p(y2), N, Hs, Ws, Hi, Wi, Cin, _p(ax), _p(ay), relu, _p(up16), s) == 0
        torch.cuda.synchronize()
        xr = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(Hi, Wi), mode="bilinear", align_corners=True)
        pos = torch.cat([ax.t()[:, None, :].expand(Cin // 2, Hi, Wi), ay.t()[:, :, None].expand(Cin // 2, Hi, Wi)], 0)
        ref = torch.nn.functional.conv2d((xr + pos[None]).half().float(), w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
        if relu:
            ref = torch.relu(ref)
        assert torch.isfinite(y2).all()
        assert _rel(y2, ref) < 5e-5, _rel(y2, ref)   # one f16 rounding of an interpolated value may flip between the two arithmetic orders
        if not relu:
            assert _rel(y2, y1) < 5e-5
        again = y2.clone()
        assert L_.wm_op_up_conv_n32(F16, _p(x), _p(w.view(torch.int16)), _p(b), _p(y2), N, Hs, Ws, Hi, Wi, Cin, _p(ax), _p(ay), relu, _p(up16), s) == 0
        torch.cuda.synchronize()
        assert torch.equal(again, y2)


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("N,gh,gw,k,Cin", [(2, 16, 16, 4, 256), (3, 10, 7, 2, 512), (1, 37, 37, 4, 256), (8, 37, 37, 2, 512), (1, 1, 5, 4, 64), (2, 3, 1, 2, 128)])
def test_token_conv_composition(dev, dt, N, gh, gw, k, Cin):
    """wm_op_tconv: Conv2d(3x3, pad 1, no bias) o ConvTranspose2d(kernel = stride = k) as ONE block-sparse GEMM at the token resolution
    (per output phase the combined matrices of the <= 2 x 2 neighbour tokens its taps land in; gemm.hip WM_EPI_CONV tc_k, wm_model.cpp
    build_tconv) — against fp32 torch conv2d(conv_transpose2d(tokens)) on the same 16-bit tokens: interior, the image border (zero
    padding of the 3x3 sees neither neighbour tokens nor the ConvTranspose bias there), single-row / single-column token grids, both k."""
    g = torch.Generator().manual_seed(N * 7 + gh + k + Cin + dt)
    tok = _t16(torch.randn(N, gh, gw, Cin, generator=g), dt).to(dev)
    wct = torch.randn(Cin, Cin, k, k, generator=g) / math.sqrt(Cin)
    bct = torch.randn(Cin, generator=g)
    wrn = torch.randn(256, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    out = torch.full((N, k * gh, k * gw, 256), float("nan"), device=dev)
    zero = torch.zeros(128, dtype=torch.int16, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    hp = lambda t: C.c_void_p(t.data_ptr())
    assert _lib().wm_op_tconv(dt, _p(tok), hp(wct), hp(bct), hp(wrn), _p(out), N, gh, gw, k, Cin, Cin, _p(zero), s) == 0
    torch.cuda.synchronize()
    x = tok.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(torch.nn.functional.conv_transpose2d(x, wct.to(dev), bct.to(dev), stride=k), wrn.to(dev), None, padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(out).all()
    e = _rel(out, ref)
    print(f"token conv {N}x{gh}x{gw} k{k} {Cin}->256 dt{dt}: {e:.2e}")
    assert e < (6e-4 if dt == F16 else 5e-3), e      # one rounding of each combined weight to the operand type (2^-11 / 2^-8), fp32 accumulation
    b = (out - ref).abs()
    border = torch.cat([b[:, 0].flatten(), b[:, -1].flatten(), b[:, :, 0].flatten(), b[:, :, -1].flatten()])
    assert border.max() <= 4 * b.max().clamp_min(1e-6) and border.mean() < 3 * b.mean() + 1e-6   # the border is no worse than the interior


@pytest.mark.parametrize("N,Hi,Wi,Ho,Wo,Cin,Co", [(2, 20, 16, 40, 32, 256, 128), (1, 9, 11, 33, 40, 64, 64), (3, 37, 37, 74, 74, 256, 128),
                                                  (1, 148, 148, 296, 296, 256, 128), (2, 12, 10, 12, 10, 128, 32), (1, 5, 7, 1, 1, 64, 32)])
def test_upconv3x3_tap_form(dev, N, Hi, Wi, Ho, Wo, Cin, Co):
    """wm_op_upconv3x3_tap: conv3x3(F.interpolate(x, align_corners=True)) as nine 1x1 products at the low resolution (one GEMM) + a bilinear
    gather (upconv.hip; conv's channel mixing commutes with the per-channel interpolation) — against fp32 torch on the same f16 x and
    weights (tolerance: the f16 rounding of the nine products), and against the fused-resize halo kernel the forward used before
    (wm_op_conv3x3_up); zero padding at the high-resolution border, non-integer and identity scales, a 1x1 output, bit-exact repeat."""
    g = torch.Generator().manual_seed(N * 100 + Hi + Cin + Co)
    x = torch.randn(N, Hi, Wi, Cin, generator=g).half().to(dev)
    w = (torch.randn(Co, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).half().to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    out = torch.full((N, Ho, Wo, Co), float("nan"), device=dev)
    wt = torch.empty(9 * Co * Cin, dtype=torch.int16, device=dev)
    y16 = torch.empty(N * Hi * Wi * 9 * Co, dtype=torch.int16, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_ = _lib()
    assert L_.wm_op_upconv3x3_tap(F16, _p(x), _p(w), _p(b), _p(out), N, Hi, Wi, Ho, Wo, Cin, Co, _p(wt), _p(y16), s) == 0
    torch.cuda.synchronize()
    xr = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=True)
    ref = torch.nn.functional.conv2d(xr, w.float().permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(out).all()
    e = _rel(out, ref)
    print(f"upconv tap form {N}x{Hi}x{Wi}->{Ho}x{Wo} {Cin}->{Co}: {e:.2e}")
    assert e < 4e-4, e          # nine f16-rounded partial products per output (2^-11 each), fp32 everything else
    if Hi * Wi >= 256 and Ho * Wo >= 256 and Co % 4 == 0:
        y1 = torch.empty(N, Ho, Wo, Co, device=dev)
        assert L_.wm_op_conv3x3_up(F16, _p(x.float()), _p(w), _p(b), _p(y1), N, Hi, Wi, Ho, Wo, Cin, Co, None, None, s) == 0
        torch.cuda.synchronize()
        assert _rel(out, y1) < 6e-4, _rel(out, y1)   # the halo kernel rounds the INTERPOLATED values to f16 instead
    again = out.clone()
    assert L_.wm_op_upconv3x3_tap(F16, _p(x), _p(w), _p(b), _p(out), N, Hi, Wi, Ho, Wo, Cin, Co, _p(wt), _p(y16), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(again, out)


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("N,Hh,Ww,Cin,Cout", [(2, 37, 37, 256, 256), (1, 64, 64, 256, 256), (3, 20, 16, 64, 64), (2, 74, 74, 256, 256),
                                              (1, 33, 40, 128, 256), (2, 148, 148, 256, 256), (1, 18, 14, 256, 128)])
def test_conv3x3_as_pingpong_gemm_on_16bit_input(dev, dt, N, Hh, Ww, Cin, Cout):
    """wm_op_conv3x3_gemm16: the 3x3 conv of a 16-bit NHWC tensor as the ping-pong GEMM itself (rows = pixels, K = (tap, channel), the A
    pieces DMA-ed from the tap-shifted pixel or the zero page) — against fp32 torch on the same rounded operands: plain, with the
    ResidualConvUnit's relu(resid) + fusion add (dense_head.py:435-455), with ReLU and a 16-bit output; ragged last row band, images
    narrower than a tile row, several images per tile, row-band schedule on and off (bit-identical), bit-exact repeat."""
    g = torch.Generator().manual_seed(N * 1000 + Hh + Cin + dt)
    x16 = _t16(torch.randn(N, Hh, Ww, Cin, generator=g), dt).to(dev)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)
    w16 = _t16(w, dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    resid = torch.randn(N, Hh, Ww, Cout, generator=g).to(dev)
    resid2 = torch.randn(N, Hh, Ww, Cout, generator=g).to(dev)
    zero = torch.zeros(128, dtype=torch.int16, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xf, wf = x16.float(), w16.float()
    base = torch.nn.functional.conv2d(xf.permute(0, 3, 1, 2), wf.permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    L_ = _lib()
    for use_res, relu_out, out16 in ((False, 0, 0), (True, 0, 0), (False, 1, 1), (True, 1, 0)):
        ref = base + (torch.relu(resid) + resid2 if use_res else 0)
        if relu_out:
            ref = torch.relu(ref)
        outs = []
        for sched in (1, 0, 1):
            L_.wm_set_tuning(b"gemm_sched", sched)
            try:
                y = torch.full((N, Hh, Ww, Cout), float("nan"), device=dev) if not out16 else torch.full((N, Hh, Ww, Cout), -1, dtype=torch.int16, device=dev)
                st = L_.wm_op_conv3x3_gemm16(dt, _p(x16), _p(w16), _p(b), _p(resid) if use_res else None, 1, _p(resid2) if use_res else None, _p(y),
                                             out16, relu_out, N, Hh, Ww, Cin, Cout, _p(zero), s)
                assert st == 0
                torch.cuda.synchronize()
            finally:
                L_.wm_set_tuning(b"gemm_sched", -1)
            outs.append(y)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        y = outs[0]
        if out16:
            y = _from16(y, dt)
            tol = 2e-3 if dt == F16 else 8e-3
        else:
            tol = 2e-5
        assert torch.isfinite(y).all()
        e = _rel(y, ref)
        print(f"conv-gemm {N}x{Hh}x{Ww} {Cin}->{Cout} dt{dt} res{use_res} relu{relu_out} o16 {out16}: {e:.2e}")
        assert e < tol, e


@pytest.mark.parametrize("Cin,Cout,Hs,Ws,Hi,Wi,pos", [(256, 128, 20, 16, 40, 32, False), (128, 32, 40, 32, 70, 56, True),
                                                    (64, 64, 9, 11, 33, 40, True), (256, 128, 148, 148, 296, 296, False)])
def test_conv3x3_fused_upsample(dev, Cin, Cout, Hs, Ws, Hi, Wi, pos):
    """conv3x3(resize(x) [+ pos tables]) with the resize fused into the halo staging == torch interpolate
    (align_corners=True) + conv on 16-bit-rounded inputs, and == the unfused bilinear op followed by the conv op."""
    dt = F16
    g = torch.Generator().manual_seed(Cin + Hi)
    N = 2
    x = torch.randn(N, Hs, Ws, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    w = _t16(w, dt).float().to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    w16 = _t16(w.permute(0, 2, 3, 1).contiguous(), dt)
    addx = torch.randn(Wi, Cin // 2, generator=g).to(dev) if pos else None
    addy = torch.randn(Hi, Cin // 2, generator=g).to(dev) if pos else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.empty(N, Hi, Wi, Cout, device=dev)
    assert _lib().wm_op_conv3x3_up(dt, _p(x), _p(w16), _p(b), _p(y), N, Hs, Ws, Hi, Wi, Cin, Cout, _p(addx) if pos else None,
                                   _p(addy) if pos else None, s) == 0
    torch.cuda.synchronize()
    up = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(Hi, Wi), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    if pos:
        up = up + torch.cat([addx[None, None].expand(N, Hi, Wi, Cin // 2), addy[None, :, None].expand(N, Hi, Wi, Cin // 2)], -1)
    ref = _conv_ref(up.contiguous(), w, b, 1, 1, 0, None, False, None, dt)
    e = _rel(y, ref)
    print(f"fused upsample conv {Cin}->{Cout} {Hs}x{Ws}->{Hi}x{Wi} pos{pos}: {e:.2e}")
    # the interpolated value is rounded to 16 bit: a different fp32 summation order flips a few roundings
    assert e < 3e-4


def test_bilinear(dev):
    x = torch.randn(2, 19, 19, 64).to(dev)
    for (Ho, Wo) in ((37, 37), (40, 31), (38, 38)):
        y = torch.empty(2, Ho, Wo, 64, device=dev)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert _lib().wm_op_bilinear(_p(x), _p(y), 2, 19, 19, Ho, Wo, 64, s) == 0
        torch.cuda.synchronize()
        ref = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        assert _rel(y, ref) < 2e-6


# from the third shape on: the fp32-MFMA weight-streaming kernel (M <= 64 in 1, 2 or 4 row tiles of 16, N % 16 == 0, K % 1024 == 0):
# 4, 8 and 16 waves per block; (65, ...) falls back to the VALU streaming kernel
@pytest.mark.parametrize("M,N,K,ldx", [(3, 256, 9, 12), (12, 512, 256, 256), (64, 6144, 2048, 2048), (5, 9, 1024, 1024),
                                       (8, 6144, 2048, 2048), (13, 2048, 2048, 2080), (16, 4096, 1024, 1024), (1, 2048, 8192, 8192),
                                       (32, 8192, 2048, 2048), (17, 2048, 2048, 2048), (40, 2048, 8192, 8192), (33, 4096, 1024, 1040),
                                       (65, 2048, 2048, 2048)])
def test_linear_f32(dev, M, N, K, ldx):
    g = torch.Generator().manual_seed(M + N)
    X = torch.zeros(M, ldx)
    X[:, :K] = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    Xd, Wd, bd = X.to(dev), W.to(dev), b.to(dev)
    Y = torch.empty(M, N, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib().wm_op_linear_f32(_p(Xd), _p(Wd), _p(bd), _p(Y), M, N, K, ldx, 1, 2, s) == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(torch.nn.functional.silu(X[:, :K]) @ W.t() + b)
    assert _rel(Y.cpu(), ref) < 5e-6
