"""The 16-bit tile-GEMM kernels must own the whole register file of their SIMD (round 6, fourth session).

A wave that shares a SIMD with waves issuing gfx950's double-rate 16-bit MFMA (v_mfma_f32_32x32x16_bf16 / _f16) was measured to compute
VALU results wrong (profiles/r06d_concurrent_forwards_exclusions.md).  The two-waves-per-SIMD instantiations of gemm16_big_kernel and
gemm_x3_kernel therefore reserve v255 — 256 registers allocated, two waves = all 512 of the SIMD — so that no foreign wave can be placed
beside them.  This reads the kernel metadata of the BUILT library (no GPU, no compile) and fails if an edit drops the property."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_vgprs(tmp_path):
    lib = os.path.join(ROOT, "s3prl_amd", "libs3enc.so")
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(lib) and os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("built library or the LLVM binutils of the ROCm image are missing")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "libs3enc.so")
    subprocess.run([objdump, "--offloading", "libs3enc.so"], cwd=work, check=True, capture_output=True)
    out = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", f], cwd=work, check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s*\.vgpr_count:\s+(\d+)", line)
            if m and name:
                out[name] = int(m.group(1))
                name = None
    return out


def test_16bit_tile_gemm_kernels_allocate_the_whole_register_file(tmp_path):
    vg = _kernel_vgprs(tmp_path)
    tile = {k: v for k, v in vg.items() if "gemm16_big_kernel" in k}
    x3 = {k: v for k, v in vg.items() if "gemm_x3_kernel" in k}
    assert len(tile) >= 20 and len(x3) >= 2, (len(tile), len(x3))
    checked = 0
    for k, v in tile.items():
        # gemm16_big_kernel<T, WTM, ROWB, NST, WPE, WN, ...>: the fourth integer template argument is the waves per SIMD
        ints = re.findall(r"ELi(\d+)", k)
        wpe = int(ints[3])
        if wpe == 2:
            assert v == 256, f"{k}: {v} registers — a foreign wave fits beside two of these on a SIMD"
            checked += 1
        else:  # four waves per SIMD x 128 registers: full as well
            assert wpe * v == 512, (k, v)
    assert checked >= 20
    for k, v in x3.items():
        assert v == 256, (k, v)
