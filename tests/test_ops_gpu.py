"""Per-kernel parity tests on the GPU, through the C ABI (s3enc_op_*), against float64 numpy restatements."""

import ctypes as C
import zlib

import numpy as np
import pytest

from oracle import encoder_oracle as O

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "bf16": 1.2e-2, "fp16": 2e-3, "fp32x3": 4e-5}


def _torch():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _round(x, dtype):
    """Round an fp32 numpy array to the 16-bit operand type (what the kernels see)."""
    torch = _torch()
    if dtype in ("fp32", "fp32x3"):  # fp32x3 sees the unrounded operands (it splits them, it does not round them)
        return x.astype(np.float32)
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    t = t.to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    return t.float().numpy()


def _dev(x, dtype="fp32"):
    torch = _torch()
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    if dtype == "bf16":
        t = t.to(torch.bfloat16)
    elif dtype == "fp16":
        t = t.to(torch.float16)
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["plain", "conv", "edge"])
def test_gemm_variants(dtype, case, variant):
    """Every staging variant (128/64-byte K stages x register-staged / LDS-DMA) computes the same GEMM."""
    from s3prl_amd import _lib

    lib = _lib.load()
    _lib.check(lib.s3enc_set_tuning(b"gemm_variant", variant))
    try:
        test_gemm(dtype, case)
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gemm_variant", 3))


@pytest.mark.parametrize("mode", [1, 2, 4, 5, 6, 7, 8, 9, 10, 0])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["big_plain", "big_conv", "big_epilogue", "big_edge"])
def test_gemm16_big_tiles(dtype, case, mode):
    """The large-tile LDS-DMA kernels of the 16-bit modes (gemm16.hip): 256x256 (mode 1) and 128x256 (mode 2) tiles on
    shapes that span several tiles with ragged M / N edges, overlapping conv rows, batches and the full epilogue; mode 7 = the
    persistent tile loop (one workgroup per CU walks its tiles, the next tile's first K step issued before the epilogue);
    mode 0 runs the same shapes through the 128x128 kernel."""
    from s3prl_amd import _lib

    lib = _lib.load()
    _lib.check(lib.s3enc_set_tuning(b"gemm16_big", mode))
    try:
        test_gemm(dtype, case)
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gemm16_big", 3))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", ["plain", "conv", "epilogue", "edge"])
def test_gemm(dtype, case):
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(zlib.crc32(f"{dtype}/{case}".encode()))
    act, use_res, use_lim = 0, False, False
    Cc, Lin = 64, 301
    if case == "plain":
        batches, M, N, K, lda = 1, 300, 384, 256, 256
    elif case == "big_plain":
        batches, M, N, K, lda = 1, 700, 516, 320, 320
    elif case == "big_conv":
        Cc, Lin = 128, 1101
        M = (Lin - 3) // 2 + 1
        batches, N, K, lda = 3, 256, 3 * Cc, 2 * Cc
        act = 1
    elif case == "big_epilogue":
        batches, M, N, K, lda = 2, 530, 264, 128, 136
        act, use_res, use_lim = 1, True, True
    elif case == "big_edge":
        batches, M, N, K, lda = 2, 129, 132, 64, 64
    elif case == "conv":  # Conv1d(C, C, k=3, s=2) on channel-last rows: lda = 2C < K = 3C
        M = (Lin - 3) // 2 + 1
        batches, N, K, lda = 3, 64, 3 * Cc, 2 * Cc
        act = 1
    elif case == "epilogue":
        batches, M, N, K, lda = 2, 130, 136, 128, 128
        act, use_res, use_lim = 1, True, True
    else:  # ragged everything: M, N not multiples of the tile, K with a partial 128-byte stage
        batches, M, N, K, lda = 2, 77, 72, 200, 200
    if case in ("conv", "big_conv"):
        a_bs = Lin * Cc
        A = rng.standard_normal((batches, Lin * Cc)).astype(np.float32)
    else:
        a_bs = M * lda
        A = rng.standard_normal((batches, M * lda)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((batches, M, N)).astype(np.float32)
    lim = np.array([M - 5, M // 2, M][:batches] + [M] * max(0, batches - 3), dtype=np.int32)[:batches]

    Ar, Wr = _round(A, dtype).astype(np.float64), _round(W, dtype).astype(np.float64)
    ref = np.empty((batches, M, N))
    for b in range(batches):
        rows = np.stack([Ar[b, m * lda:m * lda + K] for m in range(M)])
        y = rows @ Wr.T + bias
        if act:
            y = O.gelu(y)
        if use_res:
            y = y + res[b]
        if use_lim:
            y[lim[b]:] = 0
        ref[b] = y

    dA, dW = _dev(A, dtype), _dev(W, dtype)
    dbias, dres, dlim = _dev(bias), _dev(res), torch.from_numpy(lim).cuda()
    out32 = torch.full((batches, M, N), float("nan"), device="cuda")
    out16 = None
    if dtype != "fp32":
        out16 = torch.zeros((batches, M, N), device="cuda", dtype=torch.bfloat16 if dtype == "bf16" else torch.float16)
    rc = lib.s3enc_op_gemm(_lib.DTYPES[dtype], _ptr(dA), lda, a_bs, _ptr(dW), _ptr(dbias), M, N, K, batches, act,
                           _ptr(dres) if use_res else None, _ptr(dlim) if use_lim else None, _ptr(out32), _ptr(out16),
                           N, M * N, None)
    _lib.check(rc, "s3enc_op_gemm")
    torch.cuda.synchronize()
    got = out32.cpu().numpy()
    assert np.isfinite(got).all()
    err = O.rel_err(got, ref)
    assert err < TOL[dtype], f"gemm {dtype}/{case}: rel-err {err:.3e}"
    # element-wise check catches a transposed / permuted tile that a norm would also catch but names the spot
    bad = np.abs(got - ref) > 50 * TOL[dtype] * (1 + np.abs(ref))
    assert not bad.any(), f"{bad.sum()} elements off, first at {np.argwhere(bad)[0]}"
    if out16 is not None:
        assert O.rel_err(out16.float().cpu().numpy(), ref) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C_", [64, 512, 768, 1024])
def test_layernorm(dtype, C_):
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(C_)
    rows = 37
    x = (rng.standard_normal((rows, C_)) * 3 + 1.5).astype(np.float32)
    g = (1 + 0.2 * rng.standard_normal(C_)).astype(np.float32)
    b = (0.3 * rng.standard_normal(C_)).astype(np.float32)
    for act in (0, 1):
        ref = O.layer_norm(x.astype(np.float64), g.astype(np.float64), b.astype(np.float64))
        if act:
            ref = O.gelu(ref)
        out32 = torch.empty((rows, C_), device="cuda")
        out16 = torch.empty((rows, C_), device="cuda", dtype=torch.bfloat16) if dtype == "bf16" else None
        dx, dg, db = _dev(x), _dev(g), _dev(b)  # keep the device tensors alive across the call
        rc = lib.s3enc_op_layernorm(_lib.DTYPES[dtype], _ptr(dx), _ptr(dg), _ptr(db), rows, C_, act,
                                    _ptr(out32), _ptr(out16), None)
        _lib.check(rc, "s3enc_op_layernorm")
        torch.cuda.synchronize()
        assert O.rel_err(out32.cpu().numpy(), ref) < 2e-6
        if out16 is not None:
            assert O.rel_err(out16.float().cpu().numpy(), ref) < 5e-3


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C_", [512, 768, 1024])
def test_layernorm_two_rows_per_wave_is_bit_identical(dtype, C_):
    """Round 6 (second session), tuning key `ln_rows` (default 1: measured 2-3 % slower, kept as an opt-in): launches of >= 8192 rows give each wave TWO rows, both rows' loads in
    flight before the first reduction — per-row arithmetic untouched, so every output bit must equal the one-row form; odd row count (the
    last wave owns one row), out of place and IN PLACE (the encoder's post-LN layers: LayerNorm 1 overwrites its input)."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(C_ + 5)
    rows = 8192 + 37
    x = (rng.standard_normal((rows, C_)) * 3 + 1.5).astype(np.float32)
    g = (1 + 0.2 * rng.standard_normal(C_)).astype(np.float32)
    b = (0.3 * rng.standard_normal(C_)).astype(np.float32)
    dg, db = _dev(g), _dev(b)
    res = {}
    try:
        for r in (1, 2):
            _lib.check(lib.s3enc_set_tuning(b"ln_rows", r))
            for inplace in (False, True):
                dx = _dev(x)
                out32 = dx if inplace else torch.full((rows, C_), float("nan"), device="cuda")
                out16 = torch.zeros((rows, C_), device="cuda", dtype=torch.bfloat16) if dtype == "bf16" else None
                _lib.check(lib.s3enc_op_layernorm(_lib.DTYPES[dtype], _ptr(dx), _ptr(dg), _ptr(db), rows, C_, 0, _ptr(out32), _ptr(out16),
                                                  None), "s3enc_op_layernorm")
                torch.cuda.synchronize()
                res[(r, inplace)] = (out32.view(torch.int32).cpu().numpy().copy(),
                                     None if out16 is None else out16.view(torch.int16).cpu().numpy().copy())
    finally:
        _lib.check(lib.s3enc_set_tuning(b"ln_rows", 1))
    for inplace in (False, True):
        assert np.array_equal(res[(1, False)][0], res[(2, inplace)][0])
        assert np.array_equal(res[(1, False)][0], res[(1, inplace)][0])
        if dtype == "bf16":
            assert np.array_equal(res[(1, False)][1], res[(2, inplace)][1])


def _attention_ref(qkv, valid, B, T, H, table=None, gate=None, R=None):
    D = H * 64
    q = qkv[:, :D].reshape(B, T, H, 64).transpose(0, 2, 1, 3)
    k = qkv[:, D:2 * D].reshape(B, T, H, 64).transpose(0, 2, 1, 3)
    v = qkv[:, 2 * D:].reshape(B, T, H, 64).transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2)
    if table is not None:
        idx = np.clip(np.arange(T)[None, :] - np.arange(T)[:, None], -R, R) + R  # [i][j] -> clamp(j - i) + R
        bias = table[:, idx]  # (H,T,T)
        gt = gate if gate is not None else np.ones((B, H, T))
        s = s + gt[..., None] * bias[None]
    for b in range(B):
        s[b, :, :, valid[b]:] = -np.inf
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(B * T, D)


@pytest.mark.parametrize("shape", [(1, 700, 516, 320, 1, True), (2, 530, 264, 128, 0, False), (1, 129, 132, 64, 0, False),
                                   (1, 1000, 768, 3072, 1, False)])
def test_gemm_f16x2_two_term_weights(shape):
    """S3ENC_F16X2 at the op level: fp16 A, W as [hi | lo] fp16 halves per row, the K loop runs over 2K with A read twice.
    With A already fp16-exact the result must match the float64 product with the UNROUNDED weights to fp32-accumulation
    level — the weights' fp16 rounding error (5e-4 relative per weight) is gone — on the large-tile kernel and on the
    128x128 fallback (small shapes), with GELU / residual epilogues."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    batches, M, N, K, act, use_res = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    A = _round(rng.standard_normal((batches, M, K)).astype(np.float32), "fp16")
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((batches, M, N)).astype(np.float32)
    y = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if act:
        y = O.gelu(y)
    if use_res:
        y = y + res
    Wt = torch.from_numpy(W)
    hi = Wt.half()
    lo = (Wt - hi.float()).half()
    W2 = torch.cat([hi, lo], dim=1).contiguous().cuda()  # (N, 2K)
    dA, dbias, dres = _dev(A, "fp16"), _dev(bias), _dev(res)
    out = torch.full((batches, M, N), float("nan"), device="cuda")
    rc = lib.s3enc_op_gemm(4, _ptr(dA), K, M * K, _ptr(W2), _ptr(dbias), M, N, K, batches, act, _ptr(dres) if use_res else None, None,
                           _ptr(out), None, N, M * N, None)
    _lib.check(rc, "s3enc_op_gemm f16x2")
    torch.cuda.synchronize()
    err = O.rel_err(out.cpu().numpy(), y)
    assert err < 2e-6, f"f16x2 gemm {shape}: rel-err {err:.3e}"
    # the same A against the fp16-ROUNDED weights alone (what S3ENC_F16 computes) is two orders worse
    out1 = torch.full((batches, M, N), float("nan"), device="cuda")
    dW1 = hi.contiguous().cuda()
    _lib.check(lib.s3enc_op_gemm(2, _ptr(dA), K, M * K, _ptr(dW1), _ptr(dbias), M, N, K, batches, act,
                                 _ptr(dres) if use_res else None, None, _ptr(out1), None, N, M * N, None), "s3enc_op_gemm f16")
    torch.cuda.synchronize()
    assert O.rel_err(out1.cpu().numpy(), y) > 20 * err


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 4000, 2304, 768, 0, False, True), (1, 3000, 3072, 768, 1, False, True), (2, 1500, 768, 3072, 0, True, False),
                                   (1, 9000, 512, 1536, 1, False, False), (1, 99, 768, 768, 0, False, True), (3, 333, 520, 256, 1, False, True),
                                   (1, 700, 1024, 128, 0, True, False)])
def test_gemm_f16x2_mx_second_term(shape):
    """Round 5: S3ENC_F16X2's lo weight term as an MX-fp4 image on the scaled-MFMA pipe (gemm16.hip MXW; op code 5 packs the images
    exactly as s3enc_create does).  Against the float64 product with the UNROUNDED weights, on fp16-exact activations with outlier
    channels and heavy-tailed weights: the MX term must remove most of the one-term weight error (measured 4.8e-5 vs 2.2e-4,
    profiles/r04_mx_gemm_lab.md; two fp16 terms: 5e-7) on every epilogue the mode uses — 16-bit out plain / GELU (row-per-lane
    form), fp32 out + residual, fp32 out + GELU — on ragged edges, batches and a 99-row product; and be run-to-run bit-identical."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    batches, M, N, K, act, use_res, out16 = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    A = rng.standard_normal((batches, M, K)).astype(np.float32)
    A[..., ::193] *= 50.0                                                   # a few outlier channels
    A = _round(A, "fp16")
    W = (rng.standard_t(4.0, size=(N, K)) / np.sqrt(2.0 * K)).astype(np.float32)  # heavy-tailed
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((batches, M, N)).astype(np.float32)

    def ref(Wm):
        y = A.astype(np.float64) @ Wm.astype(np.float64).T + bias
        if act:
            y = O.gelu(y)
        return y + res if use_res else y

    y = ref(W)
    y_hi = ref(W.astype(np.float16).astype(np.float32))                    # what one fp16 term computes
    one_term = O.rel_err(y_hi, y)
    dA, dW, dbias, dres = _dev(A, "fp16"), _dev(W), _dev(bias), _dev(res)

    def run():
        o32 = None if out16 else torch.full((batches, M, N), float("nan"), device="cuda")
        o16 = torch.full((batches, M, N), float("nan"), device="cuda").half() if out16 else None
        rc = lib.s3enc_op_gemm(5, _ptr(dA), K, M * K, _ptr(dW), _ptr(dbias), M, N, K, batches, act, _ptr(dres) if use_res else None, None,
                               _ptr(o32) if o32 is not None else None, _ptr(o16) if o16 is not None else None, N, M * N, None)
        _lib.check(rc, "s3enc_op_gemm f16x2 + mx")
        torch.cuda.synchronize()
        return o16 if out16 else o32

    out = run()
    assert torch.equal(out, run()), "not run-to-run bit-identical"
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    if out16:  # the output's own fp16 rounding (2.8e-4) hides the weights' error: a loose bound on the fp16 grid (the fp32-output
               # shapes carry the accuracy claim)
        err = O.rel_err(got, y.astype(np.float16).astype(np.float64))
        assert err < 3e-4, f"mx second term {shape}: rel-err on the fp16 grid {err:.3e}"
    else:
        err = O.rel_err(got, y)
        assert err < 8e-5 and err < 0.35 * one_term, f"mx second term {shape}: rel-err {err:.3e}, one fp16 term {one_term:.3e}"


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16", "fp32x3"])
@pytest.mark.parametrize("T,rel", [(33, False), (200, False), (149, 200), (499, False), (300, 40), (749, 800)])
def test_attention(dtype, T, rel):
    """``rel``: False, or the half-width R of the (H, 2R+1) relative-position table (R < T-1 exercises the clamp where
    the WavLM bucket saturates, R >= T-1 the unclamped window)."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(T)
    B, H = 3, 2
    D = 64 * H
    qkv = rng.standard_normal((B * T, 3 * D)).astype(np.float32)
    qkv[:, :D] *= 0.35  # q arrives pre-scaled; keep logits O(few)
    # one outlier key per batch row exercises the online-softmax rescale
    qkv[T // 2, D:2 * D] *= 4.0
    valid = np.array([T, max(1, T // 3), max(1, T - 7)], dtype=np.int32)
    table = gate = None
    if rel:
        table = rng.standard_normal((H, 2 * rel + 1)).astype(np.float32)
        gate = (1 + rng.random((B, H, T))).astype(np.float32)
    qr = _round(qkv, dtype).astype(np.float64)
    qdev = qkv
    if dtype in ("bf16", "fp16"):
        # the 16-bit kernels take q pre-scaled by log2(e) as well (base-2 scores, s3enc.h): round q * log2(e) to the operand
        # type once, and let the reference see exactly that operand divided by log2(e)
        log2e = 1.4426950408889634
        qdev = qkv.copy()
        qdev[:, :D] = _round(qkv[:, :D] * np.float32(log2e), dtype)
        qr[:, :D] = qdev[:, :D].astype(np.float64) / log2e
    ref = _attention_ref(qr, valid, B, T, H, None if table is None else table.astype(np.float64),
                         None if gate is None else gate.astype(np.float64), R=rel)
    dq = _dev(qdev, dtype)
    out = torch.zeros((B * T, D), device="cuda", dtype=dq.dtype)
    dvalid = torch.from_numpy(valid).cuda()
    dtable = _dev(table) if rel else None  # keep the device tensors alive across the call
    dgate = _dev(gate) if rel else None
    rc = lib.s3enc_op_attention(_lib.DTYPES[dtype], _ptr(dq), _ptr(out), _ptr(dvalid), B, T, H, _ptr(dtable), int(rel),
                                _ptr(dgate), None)
    _lib.check(rc, "s3enc_op_attention")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    err = O.rel_err(got, ref)
    assert err < {"fp32": 2e-5, "bf16": 1.5e-2, "fp16": 2e-3, "fp32x3": 5e-5}[dtype], f"attention {dtype} T={T}: rel-err {err:.3e}"


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("T,rel", [(37, False), (499, False), (300, 40), (749, 800)])
def test_persistent_attention_is_bit_identical_to_the_one_shot_grid(dtype, T, rel):
    """Round 6, tuning key `attn_persist` (default 0: a measured prototype that lost, profiles/r06_attention_persist.md): persistent
    workgroups walk the (batch, head, query block) items and fetch the next item's operands through a buffer descriptor (rows past
    the last frame read as zero instead of re-reading it) — every product and every softmax update in the same order, so the
    result must equal the one-shot grid's bit for bit, ragged batches, an utterance of ONE valid frame and the WavLM bias included."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(T + 1)
    B, H = 5, 3  # 15 (batch, head) units over 8 XCDs: uneven item lists, some workgroups walk several items
    D = 64 * H
    qkv = rng.standard_normal((B * T, 3 * D)).astype(np.float32)
    qkv[:, :D] *= 0.35
    qkv[T // 2, D:2 * D] *= 4.0
    valid = np.array([T, max(1, T // 3), max(1, T - 7), 1, max(1, T - 64)], dtype=np.int32)
    table = gate = None
    if rel:
        table = rng.standard_normal((H, 2 * rel + 1)).astype(np.float32)
        gate = (1 + rng.random((B, H, T))).astype(np.float32)
    dq = _dev(qkv, dtype)
    dvalid = torch.from_numpy(valid).cuda()
    dtable = _dev(table) if rel else None
    dgate = _dev(gate) if rel else None
    outs = []
    try:
        for persist in (0, 1):
            _lib.check(lib.s3enc_set_tuning(b"attn_persist", persist))
            out = torch.full((B * T, D), float("nan"), device="cuda", dtype=dq.dtype)
            _lib.check(lib.s3enc_op_attention(_lib.DTYPES[dtype], _ptr(dq), _ptr(out), _ptr(dvalid), B, T, H, _ptr(dtable), int(rel),
                                              _ptr(dgate), None), "s3enc_op_attention")
            torch.cuda.synchronize()
            outs.append(out.view(torch.int16 if dq.dtype != torch.float32 else torch.int32).cpu().numpy())
    finally:
        _lib.check(lib.s3enc_set_tuning(b"attn_persist", 0))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16", "fp32x3"])
def test_attention_with_every_score_far_below_zero(dtype):
    """q . b_k shifts all scores of a query by the same amount — softmax-invariant, so nothing in training bounds it.  With
    every score near -320 (log2 domain: -460) the first rescale of the 16-bit kernel would compute exp2(+460) = inf and
    0 * inf = NaN for the whole row; the reference (and the result) is an ordinary softmax of the small differences."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)
    B, T, H = 2, 150, 2
    D = 64 * H
    qkv = (0.1 * rng.standard_normal((B * T, 3 * D))).astype(np.float32)
    qkv[:, :D] += 1.0
    qkv[:, D:2 * D] -= 5.0
    qkv[:, 2 * D:] = rng.standard_normal((B * T, D)).astype(np.float32)
    valid = np.array([T, 70], dtype=np.int32)
    qr = _round(qkv, dtype).astype(np.float64)
    qdev = qkv
    if dtype in ("bf16", "fp16"):
        log2e = 1.4426950408889634
        qdev = qkv.copy()
        qdev[:, :D] = _round(qkv[:, :D] * np.float32(log2e), dtype)
        qr[:, :D] = qdev[:, :D].astype(np.float64) / log2e
    ref = _attention_ref(qr, valid, B, T, H, None, None, R=False)
    dq = _dev(qdev, dtype)
    out = torch.zeros((B * T, D), device="cuda", dtype=dq.dtype)
    dvalid = torch.from_numpy(valid).cuda()
    _lib.check(lib.s3enc_op_attention(_lib.DTYPES[dtype], _ptr(dq), _ptr(out), _ptr(dvalid), B, T, H, None, 0, None, None),
               "s3enc_op_attention")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all(), "NaN / inf rows: the first rescale multiplied 0 by exp2(+huge)"
    err = O.rel_err(got, ref)
    # the scores are ~320 in magnitude: their fp32 / 16-bit-operand rounding (not the kernel's bookkeeping) sets the error
    assert err < {"fp32": 2e-4, "bf16": 6e-2, "fp16": 1e-2, "fp32x3": 5e-4}[dtype], f"{dtype}: rel-err {err:.3e}"


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16", "fp32x3"])
@pytest.mark.parametrize("shape", [(2, 499, 768, 16, 128), (1, 300, 1024, 16, 128), (3, 70, 128, 4, 16), (2, 257, 768, 16, 128)])
def test_posconv(dtype, shape):
    """Positional conv + GELU + residual (posconv.hip: fp32 Toeplitz kernel and the 16-bit implicit-GEMM kernel) against
    the oracle's pos_conv on the same (operand-rounded) inputs; Dg = 48 / 64 / 32, tiles with ragged frame counts."""
    torch = _torch()
    from s3prl_amd import _lib
    from types import SimpleNamespace

    lib = _lib.load()
    B, T, D, G, K = shape
    Dg = D // G
    rng = np.random.default_rng(zlib.crc32(f"pc/{dtype}/{shape}".encode()))
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    w = (rng.standard_normal((D, Dg, K)) / np.sqrt(Dg * K)).astype(np.float32) * 3.0
    bias = rng.standard_normal(D).astype(np.float32)
    # reference in float64 on the operands the kernel sees (16-bit modes round x and w, not the residual)
    cfg = SimpleNamespace(conv_pos=K, conv_pos_groups=G)
    xr, wr = _round(x, dtype).astype(np.float64), _round(w, dtype).astype(np.float64)
    W = {"encoder.pos_conv.0.weight_g": np.sqrt((wr ** 2).sum(axis=(0, 1), keepdims=True)),
         "encoder.pos_conv.0.weight_v": wr, "encoder.pos_conv.0.bias": bias.astype(np.float64)}
    ref = x.astype(np.float64) + O.pos_conv(cfg, W, xr)
    dx, db = _dev(x), _dev(bias)
    out = torch.full((B, T, D), float("nan"), device="cuda")
    wh = np.ascontiguousarray(w)
    rc = lib.s3enc_op_posconv(_lib.DTYPES[dtype], _ptr(dx), wh.ctypes.data_as(C.c_void_p), _ptr(db), B, T, D, G, K, _ptr(out), None)
    _lib.check(rc, "s3enc_op_posconv")
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    err = O.rel_err(got - x, ref - x)  # error of the conv branch itself, not hidden behind the residual
    assert err < TOL[dtype], f"posconv {dtype}/{shape}: rel-err {err:.3e}"


@pytest.mark.parametrize("tile", [0, 1, 2, 4])
@pytest.mark.parametrize("case", ["big_plain", "big_conv", "big_epilogue", "x3_long"])
def test_gemm_x3_split_precision(case, tile):
    """gemm_x3.hip: fp32 operands, every product rebuilt from three bf16 MFMAs (hi*hi + lo*hi + hi*lo).  Against the
    float64 product of the UNROUNDED fp32 operands the error must sit at the 1e-5 level (vs 3e-3 for plain bf16 and
    1e-6 for the exact kernel), on multi-tile shapes with ragged edges, overlapping conv rows and the full epilogue."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    # 0: gemm_x3.hip's lock-step 256x256 tile; 1: the default choice; 2 / 4: the 256 / 128-row tile kernel of gemmt.hip
    _lib.check(lib.s3enc_set_tuning(b"gemm_x3_tile", tile))
    rng = np.random.default_rng(zlib.crc32(f"x3/{case}".encode()))
    act, use_res, use_lim = 0, False, False
    Cc, Lin = 128, 1101
    if case == "big_plain":
        batches, M, N, K, lda = 1, 700, 516, 320, 320
    elif case == "big_conv":
        M = (Lin - 3) // 2 + 1
        batches, N, K, lda = 3, 256, 3 * Cc, 2 * Cc
        act = 1
    elif case == "big_epilogue":
        batches, M, N, K, lda = 2, 530, 264, 128, 136
        act, use_res, use_lim = 1, True, True
    else:
        batches, M, N, K, lda = 1, 1000, 768, 3072, 3072
    a_bs = Lin * Cc if case == "big_conv" else M * lda
    A = rng.standard_normal((batches, a_bs)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((batches, M, N)).astype(np.float32)
    lim = np.array([M - 5, M // 2, M][:batches], dtype=np.int32)
    ref = np.empty((batches, M, N))
    lin = np.empty((batches, M, N))
    for b in range(batches):
        rows = np.stack([A[b, m * lda:m * lda + K] for m in range(M)]).astype(np.float64)
        y = rows @ W.astype(np.float64).T + bias
        lin[b] = y
        if act:
            y = O.gelu(y)
        if use_res:
            y = y + res[b]
        if use_lim:
            y[lim[b]:] = 0
        ref[b] = y
    dA, dW, dbias, dres, dlim = _dev(A), _dev(W), _dev(bias), _dev(res), torch.from_numpy(lim).cuda()
    out = torch.full((batches, M, N), float("nan"), device="cuda")
    rc = lib.s3enc_op_gemm(3, _ptr(dA), lda, a_bs, _ptr(dW), _ptr(dbias), M, N, K, batches, act,
                           _ptr(dres) if use_res else None, _ptr(dlim) if use_lim else None, _ptr(out), None, N, M * N, None)
    _lib.check(lib.s3enc_set_tuning(b"gemm_x3_tile", 1))
    _lib.check(rc, "s3enc_op_gemm x3")
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    err = O.rel_err(got, ref)
    assert err < 3e-5, f"x3 gemm {case}: rel-err {err:.3e}"
    bad = np.abs(got - ref) > 3e-4 * (1 + np.abs(ref))
    assert not bad.any(), f"{bad.sum()} elements off, first at {np.argwhere(bad)[0]}"


# ---- conv0 + GroupNorm-from-lag-sums / LayerNorm (frontend.hip) ---------------------------------------------------------
def _conv0_ref(wavs, n_max, normalize, w0, bias, gn, ln, stride):
    """float64: per-utterance layer-norm -> zero pad -> Conv1d(1, C, 10, stride) -> GroupNorm(C, C) over ALL frames incl.
    the padding | LayerNorm(C) -> GELU (wav2vec2_model.py:2879-2906)."""
    B = len(wavs)
    pad = np.zeros((B, n_max))
    for b, w in enumerate(wavs):
        w = w.astype(np.float64)
        if normalize:
            w = O.wav_normalize(w)
        pad[b, :len(w)] = w
    y = O.conv1d_channel_last(pad[:, :, None], w0.astype(np.float64)[:, None, :], None if bias is None else bias.astype(np.float64), stride)
    if gn is not None:
        y = O.group_norm_per_channel(y, gn[0].astype(np.float64), gn[1].astype(np.float64))
    else:
        y = O.layer_norm(y, ln[0].astype(np.float64), ln[1].astype(np.float64))
    return O.gelu(y)


@pytest.mark.parametrize("normalize", [0, 1])
@pytest.mark.parametrize("stride", [5, 3])
def test_gn_stats_one_block_form_is_bit_identical(normalize, stride):
    """Round 6 (second session), tuning key `gn_lag_one_block` (default 1): the GroupNorm lag sums from ONE workgroup per (4096-frame chunk,
    utterance) over an LDS-staged window — same frames per thread in the same order, the upper triangle of R mirrored (x_j x_jj is an exact
    fp64 product), same butterfly and four-wave sum — must reproduce the k0-workgroups kernel bit for bit: conv0's GroupNorm'd output is
    compared as raw bits on utterances that span several chunks, end inside a staging window, and one that is shorter than a window."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(97 + stride + normalize)
    C_ = 512
    lens = [5 * 4096 * 2 + 1777, 5 * 4096 + 9, 5 * 1024 * 3 + 2, 333, 5 * 4096 * 2 + 1776]
    wavs = [(2.0 * rng.standard_normal(n) + 0.3).astype(np.float32) for n in lens]
    w0 = (rng.standard_normal((C_, 10)) * 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
    bt = (0.05 * rng.standard_normal(C_)).astype(np.float32)
    L0 = (max(lens) - 10) // stride + 1
    dw = [_dev(w) for w in wavs]
    ptrs = (C.c_void_p * len(dw))(*[t.data_ptr() for t in dw])
    ln_ = (C.c_int64 * len(dw))(*lens)
    dw0, dg, db = _dev(w0), _dev(g), _dev(bt)
    outs = []
    try:
        for one_block in (0, 1):
            _lib.check(lib.s3enc_set_tuning(b"gn_lag_one_block", one_block))
            out = torch.full((len(lens), L0, C_), float("nan"), device="cuda", dtype=torch.float32)
            _lib.check(lib.s3enc_op_conv0(_lib.DTYPES["fp32"], ptrs, ln_, len(lens), 0, normalize, _ptr(dw0), None, _ptr(dg), _ptr(db), None,
                                          None, C_, stride, _ptr(out), None), "s3enc_op_conv0")
            torch.cuda.synchronize()
            outs.append(out.view(torch.int32).cpu().numpy())
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gn_lag_one_block", 1))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["unit", "dc_offset", "pcm_int16", "ragged_norm", "layer_norm", "layer_norm_bias"])
def test_conv0_groupnorm_layernorm(dtype, case):
    """conv0_kernel + gn_lag_kernel / gn_final_kernel directly: the GroupNorm statistics come from fp64 lag sums of the
    WAVEFORM (conv0 is linear), so a large DC offset (catastrophic cancellation in a naive E[x^2]-E[x]^2) and
    un-normalised int16-scale PCM are the cases that would break a careless implementation."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(zlib.crc32(f"conv0/{case}".encode()))
    C_, stride = 512, 5
    lens = [4000, 2345, 3111] if case != "unit" else [3200, 3200]
    wavs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    normalize = False
    if case == "dc_offset":
        wavs = [(1e-2 * w + 5.0).astype(np.float32) for w in wavs]
    elif case == "pcm_int16":
        wavs = [np.round(8000.0 * w + 300.0).clip(-32768, 32767).astype(np.float32) for w in wavs]
    elif case == "ragged_norm":
        wavs = [(3.0 * w - 0.7).astype(np.float32) for w in wavs]
        normalize = True
    w0 = (rng.standard_normal((C_, 10)) * 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
    bt = (0.05 * rng.standard_normal(C_)).astype(np.float32)
    use_ln = case.startswith("layer_norm")
    bias = (0.1 * rng.standard_normal(C_)).astype(np.float32) if case == "layer_norm_bias" else None
    if use_ln:
        normalize = True
    n_max = max(lens)
    L0 = (n_max - 10) // stride + 1
    ref = _conv0_ref(wavs, n_max, normalize, w0, bias, None if use_ln else (g, bt), (g, bt) if use_ln else None, stride)
    dw = [_dev(w) for w in wavs]
    ptrs = (C.c_void_p * len(dw))(*[t.data_ptr() for t in dw])
    ln_ = (C.c_int64 * len(dw))(*lens)
    dw0, dg, db = _dev(w0), _dev(g), _dev(bt)
    dbias = _dev(bias) if bias is not None else None
    out = torch.zeros((len(lens), L0, C_), device="cuda", dtype={"fp32": torch.float32, "bf16": torch.bfloat16}[dtype])
    rc = lib.s3enc_op_conv0(_lib.DTYPES[dtype], ptrs, ln_, len(lens), 0, int(normalize), _ptr(dw0), _ptr(dbias),
                            None if use_ln else _ptr(dg), None if use_ln else _ptr(db), _ptr(dg) if use_ln else None,
                            _ptr(db) if use_ln else None, C_, stride, _ptr(out), None)
    _lib.check(rc, "s3enc_op_conv0")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    err = O.rel_err(got, ref)
    # fp32: conv + norm in fp32 with fp64 statistics; bf16: only the output is rounded (2^-9 relative per element)
    tol = {"fp32": 2e-5, "bf16": 4e-3}[dtype]
    if case == "dc_offset":
        # signal std 0.015 on a conv output of magnitude ~8: the fp32 conv products round at 8 * 2^-24 BEFORE the mean is
        # removed (the reference's fp32 conv1d has the same rounding), i.e. ~3e-5 of the centred signal; what this case
        # guards against is the statistics losing digits (E[y^2] - E[y]^2 in fp32 would be off by > 1e-1 here)
        tol = max(tol, 2e-4)
    assert err < tol, f"conv0 {case}/{dtype}: rel-err {err:.3e}"
    if not use_ln:  # the padded tail is part of the GroupNorm statistics AND is produced (frames of zeros -> gelu(beta'))
        b_short = int(np.argmin(lens))
        tail = got[b_short, (lens[b_short] // stride) + 2:]
        assert O.rel_err(tail, ref[b_short, (lens[b_short] // stride) + 2:]) < tol * 5 + 1e-6


def test_wavlm_gate_kernel():
    """wavlm_gate_kernel against the float64 formula of wavlm/modules.py:535-549:
    gate = a * (b * grep_a - 1) + 2 with a, b = sigmoid(sum over 4 of grep_linear(x_head))."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(77)
    for (B, T, H) in [(2, 37, 12), (3, 200, 16), (1, 5, 2)]:
        x = rng.standard_normal((B, T, H * 64)).astype(np.float32) * 1.5
        gw = (rng.standard_normal((8, 64)) * 0.2).astype(np.float32)
        gb = (rng.standard_normal(8) * 0.3).astype(np.float32)
        ga = (1 + 0.3 * rng.standard_normal(H)).astype(np.float32)
        xh = x.astype(np.float64).reshape(B, T, H, 64).transpose(0, 2, 1, 3)
        gl = (xh @ gw.astype(np.float64).T + gb).reshape(B, H, T, 2, 4).sum(-1)
        sg = 1.0 / (1.0 + np.exp(-gl))
        ref = sg[..., 0] * (sg[..., 1] * ga.astype(np.float64)[None, :, None] - 1.0) + 2.0  # (B, H, T)
        dx, dgw, dgb, dga = _dev(x), _dev(gw), _dev(gb), _dev(ga)
        out = torch.zeros((B, H, T), device="cuda")
        _lib.check(lib.s3enc_op_wavlm_gate(_ptr(dx), _ptr(dgw), _ptr(dgb), _ptr(dga), B, T, H, _ptr(out), None), "s3enc_op_wavlm_gate")
        torch.cuda.synchronize()
        err = O.rel_err(out.cpu().numpy(), ref)
        assert err < 5e-6, f"wavlm gate {B}x{T}x{H}: rel-err {err:.3e}"


@pytest.mark.parametrize("shape", [
    # (batches, M, N, K, lda, rows of a conv-style A batch or None, act, residual, row_limit)
    (1, 15968, 3072, 768, 768, None, 1, False, False),
    (1, 1000, 768, 3072, 3072, None, 0, True, False),
    (3, 3199, 512, 1536, 1024, 6399, 1, False, False),  # implicit conv: overlapping A rows, batch stride
    (2, 499, 768, 512, 512, None, 0, False, True),
    (1, 257, 256, 64, 64, None, 0, True, True),          # one row into the second 256-row tile
    (1, 193, 132, 16, 16, None, 1, True, True),          # a single K step, ragged N (clamped W rows), one row past 192
    (2, 99, 2304, 768, 768, None, 0, False, False),      # a short utterance: the 64-row tile's home shape
])
def test_gemm32_big_tile_is_bit_identical_to_the_default_kernel(shape):
    """The fp32 path's default GEMM (gemmt.hip: 256 / 192 / 128 / 64 x 128 tiles, tuning key gemm32_big = 2..5, 1 = chosen
    per shape): same instruction and per-accumulator k order as gemm_kernel<float> (gemm32_big = 0), so every tile height and
    every epilogue feature must reproduce that kernel bit for bit — a row's rounding never depends on the tile choice."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    nb, M, N, K, lda, rows, act, use_res, use_lim = shape
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(repr(shape).encode()))
    if rows is None:
        A, a_bs = torch.randn(nb * M * lda, device="cuda", generator=g), M * lda
    else:
        A, a_bs = torch.randn(nb * rows * 512, device="cuda", generator=g), rows * 512
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(nb * M * N, device="cuda", generator=g) if use_res else None
    lim = torch.tensor([M - 7 * (b + 1) for b in range(nb)], dtype=torch.int32, device="cuda") if use_lim else None
    outs = []
    try:
        for mode in (0, 1, 2, 3, 4, 5):
            _lib.check(lib.s3enc_set_tuning(b"gemm32_big", mode))
            out = torch.full((nb * M * N,), float("nan"), device="cuda")
            _lib.check(lib.s3enc_op_gemm(0, _ptr(A), lda, a_bs, _ptr(W), _ptr(bias), M, N, K, nb, act, _ptr(res) if use_res else None,
                                         _ptr(lim) if use_lim else None, _ptr(out), None, N, M * N, None))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gemm32_big", 1))
    assert torch.isfinite(outs[0]).all()
    for mode, out in enumerate(outs[1:], start=1):
        assert torch.equal(outs[0], out), f"gemm32_big = {mode} differs from the 128x128 kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [0.3, 1.0, 3.0, 12.0])
def test_one_transcendental_gelu_is_at_fp32_rounding_level(scale):
    """common.h gelu_fast: gelu(x) = max(x, 0) - |x|/2 * 2^(z Q(z)), z = |x| / sqrt 2 (one v_exp_f32, no reciprocal) — the GELU
    of every mode (the fp32 mode falls back to libm erff under the tuning key gelu32 = 0).  Through an identity
    product (W = I: every fp32 product and sum is exact) the GEMM epilogue returns gelu(A + bias): against torch's fp64
    erf-GELU (nn.GELU, wav2vec2_model.py:2896) it must sit at the level of the libm form evaluated in fp32 (~4e-8)."""
    torch = _torch()
    from s3prl_amd import _lib

    lib = _lib.load()
    M, N = 4096, 128
    g = torch.Generator(device="cuda").manual_seed(int(scale * 10))
    A = torch.randn(M, N, device="cuda", generator=g) * scale
    A[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 40.0, -40.0, 1e30, -1e30], device="cuda")
    W = torch.eye(N, device="cuda")
    ref = torch.nn.functional.gelu(A.double())
    errs = {}
    try:
        for key in (0, 1):
            _lib.check(lib.s3enc_set_tuning(b"gelu32", key))
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(lib.s3enc_op_gemm(0, _ptr(A), N, M * N, _ptr(W), None, M, N, N, 1, 1, None, None, _ptr(out), None, N, M * N, None))
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            errs[key] = ((out.double() - ref).norm() / ref.norm()).item()
            # exact at the ends: 0 -> 0, large positive -> x, large negative -> -0 (no NaN from inf * 0 or 1 + erf cancellation)
            assert out[0, 0].item() == 0.0 and out[0, 1].item() == 0.0
            assert out[0, 4].item() == 40.0 and out[0, 6].item() == A[0, 6].item() and abs(out[0, 5].item()) < 1e-30 and abs(out[0, 7].item()) < 1e-6
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gelu32", 1))
    assert errs[0] < 1.0e-7, errs            # libm erff
    assert errs[1] < 1.5e-7, errs            # the one-transcendental form: the same level


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 4500, 4100, 192, 1, False, False, True), (1, 5000, 3100, 128, 0, True, False, False),
                                   (3, 2300, 2052, 256, 1, True, True, False), (1, 16000, 2304, 768, 0, False, False, True),
                                   # full tiles with a following tile (the counted-vmcnt path of OVL), K = 64 (one K step) and
                                   # K = 128 (two), a 16-bit output whose rows are only 8-byte aligned (the uint2-store fallback)
                                   (1, 16384, 3072, 768, 1, False, False, True), (2, 8192, 1024, 64, 0, False, False, True),
                                   (1, 12288, 2048, 128, 1, False, False, True), (1, 9000, 2052, 256, 0, False, False, True),
                                   (1, 20480, 512, 1536, 1, False, False, False),
                                   # the row-per-lane epilogue (modes 9 / 10) on ragged edges: N a multiple of 8 but not of the
                                   # tile, M ragged, several batches, with and without GELU; conv-shaped (N = 512, long K)
                                   (1, 5000, 2312, 256, 1, False, False, True), (3, 1100, 520, 192, 0, False, False, True),
                                   (2, 31999, 512, 1536, 1, False, False, True)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm16_persistent_tile_loop_is_bit_identical(dtype, shape):
    """gemm16_big = 7: one workgroup per CU walks its tiles (more tiles than CUs here: 306 / 260 / 243-324 / 567-756), issuing the
    next tile's first K step before the current tile's epilogue.  Same MFMA order per accumulator as the one-tile-per-workgroup
    launch (mode 1), so every output — 16-bit and fp32, GELU, residual, padded-row zeroing, ragged edges — must be bit-identical."""
    torch = _torch()
    from s3prl_amd import _lib

    MODES, PP_DEFAULT = (1, 7, 8, 9, 10, 107, 109), 0
    lib = _lib.load()
    nb, M, N, K, act, use_res, use_lim, out16 = shape
    code = {"bf16": 1, "fp16": 2}[dtype]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(repr(shape).encode()))
    A = torch.randn(nb * M * K, device="cuda", generator=g).to(tdt)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(tdt)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(nb * M * N, device="cuda", generator=g) if use_res else None
    lim = torch.tensor([M - 300 * (b + 1) for b in range(nb)], dtype=torch.int32, device="cuda") if use_lim else None
    outs = []
    try:
        # 8: + the epilogue's stores draining under the next tile (OVL); 9: + the row-per-lane epilogue without LDS (SWAP) where the
        # epilogue is 16-bit-only; 10: both; 107 / 109 (round 5): modes 7 / 9 with gemm16_pp = 1 — the second half of the waves
        # issues its LDS-DMA pieces two fragment steps later than the first (a schedule change only)
        for mode in MODES:
            _lib.check(lib.s3enc_set_tuning(b"gemm16_big", mode % 100))
            _lib.check(lib.s3enc_set_tuning(b"gemm16_pp", mode // 100))
            o32 = None if out16 else torch.full((nb * M * N,), float("nan"), device="cuda")
            o16 = torch.full((nb * M * N,), float("nan"), device="cuda").to(tdt) if out16 else None
            _lib.check(lib.s3enc_op_gemm(code, _ptr(A), K, M * K, _ptr(W), _ptr(bias), M, N, K, nb, act, _ptr(res) if use_res else None,
                                         _ptr(lim) if use_lim else None, _ptr(o32) if o32 is not None else None,
                                         _ptr(o16) if o16 is not None else None, N, M * N, None))
            torch.cuda.synchronize()
            outs.append(o16 if out16 else o32)
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gemm16_big", 3))
        _lib.check(lib.s3enc_set_tuning(b"gemm16_pp", PP_DEFAULT))
    assert torch.isfinite(outs[0].float()).all()
    for mode, o in zip(MODES[1:], outs[1:]):
        assert torch.equal(outs[0], o), f"gemm16_big = {mode} differs from one tile per workgroup"
    # and the product itself, on a slice of rows of the first batch
    rows = slice(0, 512)
    ref = A[: M * K].view(M, K)[rows].double() @ W.double().T + bias.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    if use_res:
        ref = ref + res[: M * N].view(M, N)[rows].double()
    got = outs[1][: M * N].view(M, N)[rows].double()
    assert ((got - ref).norm() / ref.norm()).item() < (1e-2 if out16 else 1e-5)
