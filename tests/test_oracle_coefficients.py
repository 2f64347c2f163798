"""Pins the reconstruction coefficients: the reference's jldoctest vectors
(src/Advection/reconstruction_coefficients.jl:160-170) -> generator -> C oracle constants -> HIP header constants."""
import ctypes as C
import os
import re

import numpy as np

from oracle import coefficients as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jldoctest_float32_centered4():
    # :(-0.083333254f0 * ψ[i + -2] + 0.5833333f0 * ψ[i + -1] + 0.5833333f0 * ψ[i + 0] + -0.083333336f0 * ψ[i + 1])
    st = K.calc_reconstruction_stencil(np.float32, 2, "symmetric")
    assert [c for c, _ in st] == [-2, -1, 0, 1]
    expect = [np.float32(-0.083333254), np.float32(0.5833333), np.float32(0.5833333), np.float32(-0.083333336)]
    assert [v for _, v in st] == expect


def test_jldoctest_float32_left_upwind5():
    # :(0.0333333f0 * ψ[i-3] + -0.21666667f0 * ψ[i-2] + 0.78333336f0 * ψ[i-1] + 0.45f0 * ψ[i] + -0.05f0 * ψ[i+1])
    st = K.calc_reconstruction_stencil(np.float32, 3, "left")
    assert [c for c, _ in st] == [-3, -2, -1, 0, 1]
    expect = [np.float32(0.0333333), np.float32(-0.21666667), np.float32(0.78333336), np.float32(0.45), np.float32(-0.05)]
    assert [v for _, v in st] == expect


def test_jldoctest_trivial_stencils():
    assert K.calc_reconstruction_stencil(np.float64, 1, "left") == [(-1, 1.0)]
    assert K.calc_reconstruction_stencil(np.float32, 1, "right") == [(0, np.float32(1.0))]
    assert K.calc_reconstruction_stencil(np.float64, 1, "symmetric") == [(-1, 0.5), (0, 0.5)]


def test_coefficients_sum_to_one_and_are_close_to_rationals():
    for r, exact in zip(range(3), ((1 / 3, 5 / 6, -1 / 6), (-1 / 6, 5 / 6, 1 / 3), (1 / 3, -7 / 6, 11 / 6))):
        c = K.weno_coeff_p(np.float64, 3, r)
        assert abs(sum(c) - 1) <= 2e-16
        np.testing.assert_allclose(c, exact, rtol=0, atol=3e-16)


def test_c_oracle_constants_match_generator(oracle):
    c4 = (C.c_double * 4)()
    w5 = (C.c_double * 9)()
    w3 = (C.c_double * 4)()
    eps = C.c_double()
    oracle.lib().ocn_oracle_coefficients(c4, w5, w3, C.byref(eps))
    gen_c4 = [float(v) for _, v in K.calc_reconstruction_stencil(np.float64, 2, "symmetric")]
    assert list(c4) == gen_c4
    gen_w5 = [float(v) for r in range(3) for v in K.weno_coeff_p(np.float64, 3, r)]
    assert list(w5) == gen_w5
    gen_w3 = [float(v) for r in range(2) for v in K.weno_coeff_p(np.float64, 2, r)]
    assert list(w3) == gen_w3
    assert eps.value == float(np.float32(1e-8))  # const ε = 1f-8 (weno_interpolants.jl:70)


def test_hip_header_constants_match_generator():
    src = open(os.path.join(ROOT, "oceananigans.jl_amd", "csrc", "ocn_weno.h")).read()
    defs = {m.group(1): float(m.group(2)) for m in re.finditer(r"#define (OCN_\w+) \((-?[0-9.e+-]+)\)", src)}
    gen_c4 = [float(v) for _, v in K.calc_reconstruction_stencil(np.float64, 2, "symmetric")]
    assert [defs[f"OCN_C4_{i}"] for i in range(4)] == gen_c4
    for r in range(3):
        assert [defs[f"OCN_W5P_{r}{j}"] for j in range(3)] == [float(v) for v in K.weno_coeff_p(np.float64, 3, r)]
    assert defs["OCN_WENO_EPS"] == float(np.float32(1e-8))


def test_upwind_biased_constants_match_generator(oracle):
    """UpwindBiased(order=5) / (order=3) stencils: C oracle and HIP header literals == the Julia-faithful generator, whose
    Float32 order-5 left stencil is the reference's jldoctest vector (test_jldoctest_float32_left_upwind5)."""
    a, b, c, d = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_double * 3)(), (C.c_double * 3)()
    oracle.lib().ocn_oracle_upwind_coefficients(a, b, c, d)
    gen = lambda buf, sh: [float(v) for _, v in K.calc_reconstruction_stencil(np.float64, buf, sh)]
    assert list(a) == gen(3, "left") and list(b) == gen(3, "right") and list(c) == gen(2, "left") and list(d) == gen(2, "right")
    assert [o for o, _ in K.calc_reconstruction_stencil(np.float64, 3, "left")] == [-3, -2, -1, 0, 1]
    assert [o for o, _ in K.calc_reconstruction_stencil(np.float64, 3, "right")] == [-2, -1, 0, 1, 2]
    src = open(os.path.join(ROOT, "oceananigans.jl_amd", "csrc", "ocn_weno.h")).read()
    blk = src[src.index("#if OCN_UPWIND\n// UpwindBiased(order=5)"):src.index("#define weno5 weno5_nonlinear_unused")]
    lits = [float(x) for x in re.findall(r"(-?0\.\d+) \* S\d", blk)]
    assert lits == gen(3, "left") + gen(3, "right") + gen(2, "left") + gen(2, "right")
