"""Occupancy probe of the exact-fp32 GEMM: the same kernel with 5 ... 1 workgroups per CU (tuning key gemm_lds_pad: unused extra
LDS), 64- and 128-byte K stages.  usage (GPU box): python tools/gemm_occupancy_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from s3prl_amd import _lib
lib = _lib.load()
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / reps)
    return best
p = lambda t: C.c_void_p(t.data_ptr())
for name, M, N, K in (("sq8k", 8192, 8192, 8192), ("fc1", 15968, 3072, 768), ("fc2", 15968, 768, 3072)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; out = torch.empty(M, N, device="cuda")
    row = []
    for variant, stage in ((3, 64), (2, 128)):
        for pad_kib in ((0, 8, 24, 48, 100) if stage == 64 else (0, 24, 90)):
            base = 2 * 256 * stage // 1024
            wg = min(160 // (base + pad_kib), 5 if stage == 64 else 2)
            _lib.check(lib.s3enc_set_tuning(b"gemm_variant", variant)); _lib.check(lib.s3enc_set_tuning(b"gemm_lds_pad", pad_kib * 1024))
            t = bench(lambda: _lib.check(lib.s3enc_op_gemm(0, p(A), K, M * K, p(W), None, M, N, K, 1, 0, None, None, p(out), None, N, M * N, None)))
            row.append(f"{stage}B/{wg}wg:{2.0*M*N*K/t/1e9:.0f}")
    print(name, "  ".join(row), flush=True)
