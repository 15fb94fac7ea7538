"""The GEGLU epilogue's GELU (csrc/gemm_common.h: gelu_fast2) is max(x, 0) - |x| 2^P(min(|x|, 6)) with a degree-5 polynomial P fitted to
log2(erfc(a / sqrt 2) / 2).  The coefficients are read out of the header and the formula is re-evaluated here in fp32 against the exact
erf form: a typo in one constant would otherwise only show as a slightly larger parity error on the GPU."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coefficients():
    src = open(os.path.join(ROOT, "invertible_cd_amd", "csrc", "gemm_common.h")).read()
    body = src[src.index("gelu_fast2(const f32x2 x)"):]
    body = body[:body.index("return")]
    c = {}
    for name, val in re.findall(r"c(\d) = \{(-?[0-9.e-]+)f,", body):
        c[int(name)] = np.float32(val)
    assert sorted(c) == [0, 1, 2, 3, 4, 5], c
    return [c[k] for k in (5, 4, 3, 2, 1, 0)]          # Horner order


def test_gelu_polynomial_matches_the_erf_form_over_the_fp16_range():
    co = _coefficients()
    x = np.concatenate([np.linspace(-8, 8, 400001), np.linspace(-70, 70, 200001), np.array([-65504.0, 65504.0, 0.0, -0.0])]).astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(6.0))
    p = np.zeros_like(a)
    for k in co:
        p = (p * a + k).astype(np.float32)
    got = (np.maximum(x, 0) - np.abs(x) * np.exp2(p).astype(np.float32)).astype(np.float32)
    x64 = x.astype(np.float64)
    want = 0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))
    err = np.abs(got - want)
    assert err[np.abs(x) <= 70].max() < 1e-6, float(err.max())
    assert err.max() < 1e-4                                  # |x| 2^P(6) at the end of the fp16 range: 6e4 * 1e-9
    assert got[-2] == 0.0 and got[-1] == 0.0                 # gelu(+-0) = 0
    # where the result is not tiny (|gelu| >= 2^-9: the absolute bound above is then below half an fp16 ulp) the fp16 value the epilogue
    # stores differs from the exact form's by at most one ulp; in the negative tail (|gelu| ~ 1e-4 at x = -4) the 4e-7 error is a few ulps
    h_got, h_want = got.astype(np.float16), want.astype(np.float16)
    ulp = np.abs(h_got.view(np.int16).astype(np.int32) - h_want.view(np.int16).astype(np.int32))
    assert ulp[np.isfinite(h_want) & (np.abs(want) >= 2.0 ** -9)].max() <= 1
