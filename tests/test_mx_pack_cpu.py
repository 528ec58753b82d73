"""The host-side packer of S3ENC_F16X2's MX-fp4 second weight term (engine_internal.h::pack_mx4_lo, run by s3enc_create on every
q|k|v / fc1 / fc2 / conv1 weight) against a numpy restatement of the format: per row and 32-k block an E8M0 scale 2^e — the tightest
power of two with max|lo| / 2^e <= 6 — and 32 e2m1 values (0, 0.5, 1, 1.5, 2, 3, 4, 6 with a sign), element i in nibble i, each the
nearest representable one; lo = w - fp16(w).  Also the round-5 regression: a weight outside the fp16 range (hi = inf, lo = -inf)
must not send the scale search into a 2^31-step loop — it gets the format's NaN scale (0xFF) and the forward's status word reports
the non-finite product.  Built with hipcc as a host-only program (no GPU needed)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_mx4_packer_matches_the_format_and_terminates_on_out_of_range_weights(tmp_path):
    exe = str(tmp_path / "mx_pack_harness")
    build = subprocess.run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "s3prl_amd", "csrc"),
                            "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "mx_pack_harness.hip"), "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    rng = np.random.default_rng(0)
    N, K = 64, 256
    w = (rng.standard_t(4, size=(N, K)) / 16).astype(np.float32)
    w[3, 40] = 1e5          # outside the fp16 range: hi = inf, lo = -inf
    w[5, 7] = np.nan
    w[9, :32] = 0.0         # an all-zero block
    w[10, :64] = 0.5        # fp16-exact values: lo = 0
    w.tofile(str(tmp_path / "w.bin"))
    run = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "o.bin"), str(N), str(K)], capture_output=True, timeout=60)
    assert run.returncode == 0
    raw = np.fromfile(str(tmp_path / "o.bin"), dtype=np.uint8)
    kb = K // 32
    data, sc = raw[: N * kb * 16].reshape(N, kb, 16), raw[N * kb * 16:].reshape(N, kb)
    vals = np.array([0, 0.5, 1, 1.5, 2, 3, 4, 6])
    with np.errstate(all="ignore"):
        lo = (w - w.astype(np.float16).astype(np.float32)).reshape(N, kb, 32).astype(np.float64)
    nib = np.stack([data & 15, data >> 4], axis=-1).reshape(N, kb, 32)
    deq = np.where(nib & 8, -1.0, 1.0) * vals[nib & 7] * np.exp2(sc.astype(np.float64) - 127)[..., None]
    assert sc[3, 1] == 0xFF                                  # the block with the out-of-range weight: E8M0 NaN
    checked = 0
    for n in range(N):
        for b in range(kb):
            l = lo[n, b]
            if not np.isfinite(l).all():
                continue
            amax = np.abs(l).max()
            if amax == 0:
                assert (deq[n, b] == 0).all()
                continue
            s = 2.0 ** (int(sc[n, b]) - 127)
            assert 3.0 - 1e-6 < amax / s <= 6.0 + 1e-6, (n, b, amax / s)     # the tightest power-of-two scale
            grid = np.concatenate([-vals[::-1], vals]) * s
            nearest = np.abs(l[:, None] - grid[None]).min(1)
            assert (np.abs(deq[n, b] - l) <= nearest + 1e-12).all(), (n, b)  # every element is a nearest e2m1 value
            checked += 1
    assert checked >= N * kb - 6   # (2 non-finite blocks, 3 all-zero lo blocks)
