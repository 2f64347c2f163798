"""oracle/coefficients.py -- TEST INFRASTRUCTURE ONLY (see oracle/ocn_oracle.c header).

Restates `stencil_coefficients` (src/Advection/reconstruction_coefficients.jl:100-115) for
uniform grids, including the Julia evaluation details that decide the last bit:

  * `xr = xi = collect(1:100)` are Int vectors, so `numerator / denominator` is an
    Int/Int division, i.e. a *Float64* quotient (not a BigFloat one);
  * the quotient is accumulated into a `BigFloat` vector (exact for these magnitudes),
    then rounded once to `FT`;
  * the last coefficient is `1 - sum(others)` evaluated in `FT` (:112-114).

`uniform_reconstruction_coefficients` follows :134-136.  The reference's jldoctest
vectors (:160-170) are the known answers this generator is checked against in
tests/test_oracle_coefficients.py.
"""
from fractions import Fraction

import numpy as np


def _num_prod(i, m, l, r, order):
    p = 1
    for q in range(order + 1):
        if q != m and q != l:
            p *= i - (i - (r - q + 1))  # xr[i] - xi[i-(r-q+1)] with xr = xi = 1:100
    return p


def _round_to(FT, fr):
    """Correctly round the exact rational `fr` to dtype FT."""
    x = FT(float(fr))
    if FT is np.float64:
        return x
    cands = [x, np.nextafter(x, FT(np.inf)), np.nextafter(x, FT(-np.inf))]
    return min(cands, key=lambda c: abs(Fraction(float(c)) - fr))


def stencil_coefficients(FT, r, order, i=50):
    coeffs = [Fraction(0)] * order
    for j in range(order):
        for m in range(j + 1, order + 1):
            num = sum(_num_prod(i, m, l, r, order) for l in range(order + 1) if l != m)
            den = 1
            for l in range(order + 1):
                if l != m:
                    den *= (i - (r - m + 1)) - (i - (r - l + 1))
            term = (float(num) / float(den)) * float((i - (r - j)) - (i - (r - j + 1)))
            coeffs[j] += Fraction(term)
    fl = [_round_to(FT, c) for c in coeffs][:-1]
    s = fl[0]
    for v in fl[1:]:
        s = FT(s + v)
    return tuple(fl + [FT(FT(1) - s)])


def uniform_reconstruction_coefficients(FT, bias, buffer):
    if bias == "symmetric":
        return stencil_coefficients(FT, buffer - 1, 2 * buffer)
    if buffer == 1:
        return (FT(1),)
    if bias == "left":
        return stencil_coefficients(FT, buffer - 2, 2 * buffer - 1)
    if bias == "right":
        return stencil_coefficients(FT, buffer - 1, 2 * buffer - 1)
    raise ValueError(bias)


def calc_reconstruction_stencil(FT, buffer, shift):
    """Returns [(offset c, coefficient C)] as in calc_reconstruction_stencil (:173-203)."""
    N = buffer * 2
    order = N if shift == "symmetric" else N - 1
    if shift != "symmetric":
        N -= 1
    rng = list(range(1, N + 1))
    if shift == "right":
        rng = [n + 1 for n in rng]
    coeff = uniform_reconstruction_coefficients(FT, shift, buffer)
    return [(n - buffer - 1, coeff[order - idx - 1]) for idx, n in enumerate(rng)]


def weno_coeff_p(FT, buffer, stencil):
    """coeff_p for uniform directions (weno_interpolants.jl:118-119)."""
    return stencil_coefficients(FT, stencil, buffer)


if __name__ == "__main__":
    for r in range(3):
        print("W5P", r, [repr(float(v)) for v in weno_coeff_p(np.float64, 3, r)])
    for r in range(2):
        print("W3P", r, [repr(float(v)) for v in weno_coeff_p(np.float64, 2, r)])
    print("C4 applied to psi[n-2..n+1]:", [repr(float(c)) for _, c in calc_reconstruction_stencil(np.float64, 2, "symmetric")])
