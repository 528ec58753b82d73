#!/usr/bin/env python3
"""The opt-in 256x256 exact-fp32 GEMM tile (tuning key gemm32_big, gemm32big.hip) against the default 128x128 kernel:
bit-identity over the epilogue features, then timing on the shapes of the HuBERT-base forward.  usage (GPU box): python tools/gemm32_big_check.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from s3prl_amd import _lib
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def gemm(mode, A, lda, a_bs, W, bias, M, N, K, nb, act, res, lim, out):
    _lib.check(lib.s3enc_set_tuning(b"gemm32_big", mode))
    _lib.check(lib.s3enc_op_gemm(0, p(A), lda, a_bs, p(W), p(bias), M, N, K, nb, act, p(res), p(lim), p(out), None, N, M * N, None))


torch.manual_seed(0)
ok = True
for (nb, M, N, K, lda, rows, act, use_res, use_lim) in [
        (1, 15968, 3072, 768, 768, None, 1, False, False), (1, 1000, 768, 3072, 3072, None, 0, True, False),
        (3, 3199, 512, 1536, 1024, 6399, 1, False, False), (2, 499, 768, 512, 512, None, 0, False, True),
        (1, 257, 256, 64, 64, None, 0, True, True), (1, 4096, 2304, 768, 768, None, 0, False, False)]:
    if rows is None:
        A = torch.randn(nb * M * lda, device="cuda"); a_bs = M * lda
    else:
        A = torch.randn(nb * rows * 512, device="cuda"); a_bs = rows * 512
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    res = torch.randn(nb * M * N, device="cuda") if use_res else None
    lim = torch.tensor([M - 7 * (b + 1) for b in range(nb)], dtype=torch.int32, device="cuda") if use_lim else None
    o0 = torch.full((nb * M * N,), float("nan"), device="cuda"); o1 = o0.clone()
    gemm(0, A, lda, a_bs, W, bias, M, N, K, nb, act, res, lim, o0)
    gemm(2, A, lda, a_bs, W, bias, M, N, K, nb, act, res, lim, o1)
    torch.cuda.synchronize()
    same = torch.equal(o0, o1) and bool(torch.isfinite(o1).all())
    ok &= same
    print(f"bit-identical {same}: batches={nb} M={M} N={N} K={K} lda={lda} act={act} residual={use_res} row_limit={use_lim}", flush=True)


def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / reps)
    return best


print("| shape | 128x128 tile TF | 256x256 tile TF | tiles / 256 CUs |")
print("|---|---:|---:|---:|")
for name, nb, M, N, K, lda, rows in [("conv1", 32, 15999, 512, 1536, 1024, 31999), ("conv2", 32, 7999, 512, 1536, 1024, 15999),
                                      ("conv3", 32, 3999, 512, 1536, 1024, 7999), ("conv4", 32, 1999, 512, 1536, 1024, 3999),
                                      ("fc1", 1, 15968, 3072, 768, 768, None), ("qkv", 1, 15968, 2304, 768, 768, None),
                                      ("fc2", 1, 15968, 768, 3072, 3072, None), ("sq8k", 1, 8192, 8192, 8192, 8192, None)]:
    if rows is None:
        A = torch.randn(nb * M * lda, device="cuda"); a_bs = M * lda
    else:
        A = torch.randn(nb * rows * 512, device="cuda"); a_bs = rows * 512
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    out = torch.empty(nb * M * N, device="cuda")
    fl = 2.0 * nb * M * N * K
    t0 = bench(lambda: gemm(0, A, lda, a_bs, W, bias, M, N, K, nb, 1, None, None, out))
    t1 = bench(lambda: gemm(2, A, lda, a_bs, W, bias, M, N, K, nb, 1, None, None, out))
    tiles = nb * -(-M // 256) * -(-N // 256)
    print(f"| {name} ({nb} x {M} x {N} x {K}) | {fl / t0 / 1e9:.1f} | {fl / t1 / 1e9:.1f} | {tiles / 256:.2f} |", flush=True)
_lib.check(lib.s3enc_set_tuning(b"gemm32_big", 0))
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
