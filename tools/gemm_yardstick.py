#!/usr/bin/env python3
"""Yardstick, not product: the library's GEMM kernels next to torch.matmul (hipBLASLt / rocBLAS — vendor code that is NOT on the
data path) on the shapes of the HuBERT-base forward and on large squares, random and zero-filled operands.
usage (GPU box): python tools/gemm_yardstick.py > profiles/rNN_gemm_yardstick.md"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s3prl_amd import _lib

lib = _lib.load()
SHAPES = [("qkv", 15968, 2304, 768), ("out_proj", 15968, 768, 768), ("fc1", 15968, 3072, 768), ("fc2", 15968, 768, 3072),
          ("conv2 (as a plain GEMM)", 255968, 512, 1536), ("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192)]


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


print("# GEMM yardstick: libs3enc kernels vs torch.matmul (vendor BLAS, not on the data path), one MI355X")
print()
print("TFLOP/s (2MNK / time), plain GEMM without epilogue work beyond the output store; `zeros` = zero-filled operands (the DVFS probe: "
      "same instruction stream, less switching power).")
print()
print("| shape (M, N, K) | dtype | libs3enc random | torch random | libs3enc zeros | torch zeros |")
print("|---|---|---:|---:|---:|---:|")
for dtype, td in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    code = _lib.DTYPES[dtype]
    for name, M, N, K in SHAPES:
        if dtype == "fp32" and M > 100000:
            continue
        cells = []
        for fill in ("randn", "zeros"):
            A = (torch.randn(M, K, device="cuda") if fill == "randn" else torch.zeros(M, K, device="cuda")).to(td)
            W = (torch.randn(N, K, device="cuda") / K ** 0.5 if fill == "randn" else torch.zeros(N, K, device="cuda")).to(td)
            out = torch.empty(M, N, device="cuda", dtype=td)
            p = lambda t: C.c_void_p(t.data_ptr())
            o32, o16 = (p(out), None) if dtype == "fp32" else (None, p(out))

            def mine():
                _lib.check(lib.s3enc_op_gemm(code, p(A), K, M * K, p(W), None, M, N, K, 1, 0, None, None, o32, o16, N, M * N, None))

            def ref():
                torch.matmul(A, W.t(), out=out)

            fl = 2.0 * M * N * K
            cells += [fl / bench(mine) / 1e9, fl / bench(ref) / 1e9]
        print(f"| {name} ({M}, {N}, {K}) | {dtype} | {cells[0]:.0f} | {cells[1]:.0f} | {cells[2]:.0f} | {cells[3]:.0f} |", flush=True)
