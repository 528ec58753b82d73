"""The one-transcendental erf-GELU of the HIP kernels (csrc/common.h, gelu_fast), restated in numpy fp32 from the
coefficients the header defines: pins the fit (an edited coefficient fails here, without a GPU) and the claim the header and
DESIGN.md make about it — as close to an fp64 erf-GELU (nn.GELU: wav2vec2_model.py:2896, 3306) as the libm form is in fp32."""

import math
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "s3prl_amd", "csrc", "common.h")
f32 = np.float32


def _coefficients():
    text = open(HDR).read()
    q = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"#define S3_GELU_Q(\d) (-?[0-9.e+-]+)f", text)}
    assert sorted(q) == list(range(8)), q
    return np.array([q[i] for i in range(8)], dtype=f32)


def _gelu_fast(x, q):
    """Operation for operation what gelu_fast does (fp32 throughout, one exp2)."""
    x = x.astype(f32)
    ax = np.abs(x)
    z = np.minimum((ax * f32(0.70710678118654752440)).astype(f32), f32(16.0))
    p = (q[7] * z + q[6]).astype(f32)
    for k in (5, 4, 3, 2, 1, 0):
        p = (p * z + q[k]).astype(f32)
    with np.errstate(under="ignore"):
        e = np.exp2((z * p).astype(f32)).astype(f32)
    hz = (z * f32(0.70710678118654752440)).astype(f32)
    mx = ((x + ax) * f32(0.5)).astype(f32)
    return (mx.astype(np.float64) - hz.astype(np.float64) * e.astype(np.float64)).astype(f32)  # one fused multiply-add


def _erf64(x):
    return np.vectorize(math.erf)(x)


def test_coefficients_start_at_the_slope_of_erfc():
    q = _coefficients()
    # erfc(z) = 1 - 2 z / sqrt(pi) + ...  =>  Q(0) = -2 / (sqrt(pi) ln 2)
    assert abs(float(q[0]) + 2.0 / (math.sqrt(math.pi) * math.log(2.0))) < 3e-5
    assert float(q[7]) < 0.0  # Q keeps falling beyond the fitted range: 2^(z Q) -> 0, never a blow-up


def test_gelu_fast_is_at_the_rounding_level_of_the_libm_form():
    q = _coefficients()
    rng = np.random.default_rng(0)
    for scale, bound in ((0.3, 1.0e-7), (1.0, 6.0e-8), (3.0, 5.0e-8)):
        x = (rng.standard_normal(400_000) * scale).astype(f32)
        x64 = x.astype(np.float64)
        ref = 0.5 * x64 * (1.0 + _erf64(x64 / math.sqrt(2.0)))
        fast = _gelu_fast(x, q).astype(np.float64)
        libm = (f32(0.5) * x * (f32(1.0) + _erf64((x * f32(0.70710678118654752440)).astype(f32).astype(np.float64)).astype(f32))).astype(f32)
        e_fast = np.linalg.norm(fast - ref) / np.linalg.norm(ref)
        e_libm = np.linalg.norm(libm.astype(np.float64) - ref) / np.linalg.norm(ref)
        assert e_fast < bound, (scale, e_fast, e_libm)
        assert e_fast < 2.5 * e_libm, (scale, e_fast, e_libm)


def test_gelu_fast_range_and_special_values():
    q = _coefficients()
    x = np.linspace(-30.0, 30.0, 600_001)
    ref = 0.5 * x * (1.0 + _erf64(x / math.sqrt(2.0)))
    got = _gelu_fast(x, q).astype(np.float64)
    # absolute error: half an ulp of the result plus 3e-7
    assert np.all(np.abs(got - ref) <= 0.5 * np.spacing(np.abs(ref).astype(f32)).astype(np.float64) + 3e-7)
    sp = _gelu_fast(np.array([0.0, -0.0, 40.0, -40.0, 1e30, -1e30, np.inf]), q)
    assert sp[0] == 0.0 and sp[1] == 0.0 and sp[2] == 40.0 and sp[3] == 0.0 and sp[4] == f32(1e30) and sp[5] == 0.0 and sp[6] == np.inf
    with np.errstate(invalid="ignore"):
        bad = _gelu_fast(np.array([-np.inf, np.nan]), q)
    assert np.isnan(bad).all()  # as 0.5 * x * (1 + erf(x / sqrt 2)) does
