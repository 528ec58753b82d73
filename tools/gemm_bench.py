#!/usr/bin/env python3
"""A/B micro-benchmark of the GEMM kernel variants on the shapes of the HuBERT-base forward (within one process,
interleaved rounds).  usage: gemm_bench.py [dtype ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s3prl_amd import _lib

lib = _lib.load()
CHECK = True
SHAPES = {  # name: (batches, M, N, K, lda, a_bs_rows)
    "conv1": (32, 15999, 512, 1536, 1024, 31999),
    "conv2": (32, 7999, 512, 1536, 1024, 15999),
    "qkv": (1, 15968, 2304, 768, 768, None),
    "out_proj": (1, 15968, 768, 768, 768, None),
    "fc1": (1, 15968, 3072, 768, 768, None),
    "fc2": (1, 15968, 768, 3072, 3072, None),
    "ep_fc2": (1, 15968, 768, 64, 64, None),    # K = 64: prologue + epilogue only (residual, fp32 out)
    "ep_fc1": (1, 15968, 3072, 64, 64, None),   # K = 64: prologue + epilogue only (GELU, 16-bit out)
    "ep_qkv": (1, 15968, 2304, 64, 64, None),   # K = 64: prologue + epilogue only (plain, 16-bit out)
    "sq4k": (1, 4096, 4096, 4096, 4096, None),
    "sq8k": (1, 8192, 8192, 8192, 8192, None),
    "L_qkv": (1, 15968, 3072, 1024, 1024, None),
    "L_out": (1, 15968, 1024, 1024, 1024, None),
    "L_fc1": (1, 15968, 4096, 1024, 1024, None),
    "L_fc2": (1, 15968, 1024, 4096, 4096, None),
}

FILL = "randn"


def _fill(n, scale=1.0):
    """operand data: "randn" (what the path sees) or "zeros" (the DVFS probe: MI355X_MICROARCH.md 'DVFS give-back' — a
    power-limited kernel runs faster on zero-filled operands at identical instruction counts)"""
    return torch.zeros(n, device="cuda") if FILL == "zeros" else torch.randn(n, device="cuda") * scale


def run(dtype, variants=None, rounds=5, shapes=None, reps=10):
    x3 = dtype == "fp32x3"  # variants: gemm_x3_tile 0 (gemm_x3.hip) / 2..5 (gemmt.hip tile heights)
    big = dtype not in ("fp32", "fp32x3")
    if x3 and variants is None:
        variants = (0, 2, 4)
    if variants is None:
        # 16-bit modes: the large-tile kernel's configurations (gemm16_big; 0 = the 128x128 kernel); fp32: staging variants
        variants = (0, 1, 4, 5, 6) if big else (1, 3, 0, 2)
    td = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "fp32x3": torch.float32}[dtype]
    _lib.check(lib.s3enc_set_tuning(b"x3_pack_cache", 1 if x3 else 0))
    print(f"== {dtype}")
    for name, (nb, M, N, K, lda, rows) in SHAPES.items():
        if shapes and name not in shapes:
            continue
        if rows is None:
            A = _fill(M * lda).to(td); a_bs = M * lda
        else:
            A = _fill(nb * rows * 512).to(td); a_bs = rows * 512
        W = _fill(N * K, 1.0 / K ** 0.5).reshape(N, K).to(td)
        bias = torch.randn(N, device="cuda")
        # epilogue as in the encoder: conv / fc1 -> GELU, operand-type output; qkv -> plain; out_proj / fc2 -> fp32 residual
        # added, fp32 output (the residual stream)
        resid = name.split("_")[-1] in ("proj", "out", "fc2") or name.endswith("fc2")
        act = 0 if (resid or "qkv" in name) else 1
        res32 = torch.randn(nb * M * N, device="cuda") if resid else None
        out32 = torch.empty(nb * M * N, device="cuda") if (not big or resid) else None
        out16 = torch.empty(nb * M * N, device="cuda", dtype=td) if (big and not resid) else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        res = {v: [] for v in variants}
        ref = None
        for r in range(rounds + 1):
            for v in variants:
                if x3:
                    _lib.check(lib.s3enc_set_tuning(b"gemm_x3_tile", v))
                elif big:
                    _lib.check(lib.s3enc_set_tuning(b"gemm16_big", v))
                else:
                    _lib.check(lib.s3enc_set_tuning(b"gemm_variant", v))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):  # back-to-back launches: one launch alone is dominated by clock ramp / launch gaps
                    _lib.check(lib.s3enc_op_gemm(3 if x3 else _lib.DTYPES[dtype], p(A), lda, a_bs, p(W), p(bias), M, N, K, nb, act, p(res32),
                                                 None, p(out32), p(out16), N, M * N, None))
                e1.record(); torch.cuda.synchronize()
                if r: res[v].append(e0.elapsed_time(e1) / reps)
                o = (out32 if out32 is not None else out16).float()
                chk = float(o[:: 9973].double().sum())
                if ref is None and v < 100: ref = chk
                assert not CHECK or v >= 100 or abs(chk - ref) <= 1e-3 * abs(ref) + 1e-3, (name, v, chk, ref)
        fl = 2.0 * nb * M * N * K
        print(f"  {name:9s}", "  ".join(f"v{v}: {min(t):7.3f} ms {fl / min(t) / 1e9:7.1f} TF" for v, t in res.items()))

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("dtypes", nargs="*", default=["fp32", "bf16"])
    ap.add_argument("--shapes", default="")
    ap.add_argument("--variants", default="")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--fill", default="randn", choices=["randn", "zeros"], help="operand data (zeros: power / DVFS probe)")
    ap.add_argument("--probe", type=int, default=0, help="gemm_variant bits for timing probes (8: every tile loads tile 0)")
    a = ap.parse_args()
    FILL = a.fill
    if a.probe:
        _lib.check(lib.s3enc_set_tuning(b"gemm_variant", a.probe | 1))
        globals()["CHECK"] = False
    for d in a.dtypes:
        run(d, tuple(int(v) for v in a.variants.split(",")) if a.variants else None, a.rounds,
            set(a.shapes.split(",")) if a.shapes else None, a.reps)
