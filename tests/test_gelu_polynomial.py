"""CPU check of the GEGLU epilogue's GELU (csrc/igemm.hip gelu_erf): the coefficients are read from the source and the expression is
evaluated in emulated fp32 against the exact erf GELU the reference's F.gelu computes (layers/SpatialTransformer.py:37-47)."""
import os
import re

import numpy as np
from scipy.special import erf

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stable-diffusion_amd", "csrc", "igemm.hip")


def _coefficients():
    body = open(SRC).read()
    body = body[body.index("float gelu_erf(float g)"):]
    body = body[:body.index("}")]
    lead = float(re.search(r"float p = ([-+0-9.e]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"__builtin_fmaf\(p, z, ([-+0-9.e]+)f\)", body)]
    assert len(rest) == 8, "expected a degree-8 Horner chain"
    return [lead] + rest            # highest degree first


def _gelu_fp32(g):
    g = g.astype(np.float32)
    z = (np.abs(g) * np.float32(0.70710678118654752)).astype(np.float32)
    co = _coefficients()
    with np.errstate(all="ignore"):
        p = np.full_like(z, np.float32(co[0]))
        for c in co[1:]:
            p = (p * z + np.float32(c)).astype(np.float32)
        h = np.exp2(p).astype(np.float32)
        return (np.maximum(g, np.float32(0)) - np.abs(g) * h).astype(np.float32)


def test_gelu_polynomial_matches_exact_erf_gelu():
    g = np.linspace(-12.0, 12.0, 1200001).astype(np.float32).astype(np.float64)      # fp32-representable inputs
    got = _gelu_fp32(g).astype(np.float64)
    ref = g * 0.5 * (1.0 + erf(g / np.sqrt(2.0)))
    err = np.abs(got - ref)
    assert err.max() < 6e-7, err.max()                         # measured 2.7e-7 + the fp32 rounding of the result (4.8e-7 at |g| ~ 8)
    assert (err / np.maximum(np.abs(ref), 1e-3)).max() < 5e-5
    neg = g < 0
    assert np.all(got[neg] <= 0.0), "negative inputs must give non-positive outputs (sign kept)"


def test_gelu_polynomial_limits():
    big = np.array([20.0, 100.0, 1e4, 65504.0, 3e38])
    assert np.array_equal(_gelu_fp32(big), big.astype(np.float32))          # h -> 0: gelu(g) = g
    assert np.all(_gelu_fp32(-big) == 0.0)
    assert _gelu_fp32(np.array([0.0]))[0] == 0.0
