"""TEST INFRASTRUCTURE (oracle): restatement of the two pieces of Julia's *Base* library whose floating-point results leak into
the reference's grids -- a third-party dependency of the reference that is not under /root/reference (the Julia runtime itself;
`Project.toml` compat `julia = "1.9"`; the algorithm below is unchanged in base/twiceprecision.jl and base/special/trig.jl from 1.6 to
1.11):

  range(start, stop, length = n) for Float64 (Base._linspace, twiceprecision.jl): the reference builds every regular coordinate as
      `range(FT(F₋), FT(F₊), length = TF)` (src/Grids/grid_generation.jl:117-121), so x/y/z nodes are the elements of a
      StepRangeLen{Float64, TwicePrecision, TwicePrecision}, NOT c₁ + (i - 1) Δ rounded once.
  sind(x) (special/trig.jl): FPlane(latitude = φ) computes f = 2 Ω sind(φ) (src/Coriolis/f_plane.jl:38-40).

Pinned by the reference's own jldoctests, which print nodes that only this arithmetic produces (tests/test_reference_fixtures.py):
`x ∈ [3.60072e-17, 6.28319)`, `y ∈ [7.20145e-17, 12.5664)` (src/Grids/rectilinear_grid.jl:176-182) and `x ∈ [-6.90805e-17, 6.28319)`,
`y ∈ [-1.07194e-16, 3.14159)` (docs/src/grids.md:304-316).  Only tests/ may import this module.
"""
import math
import struct
from fractions import Fraction

_MAXINT_F32 = 16777216  # maxintfloat(Float32): `rat` works with the narrowed type's integer range


def _rat(x):
    """Base.rat(x): continued-fraction approximation with numerator / denominator <= maxintfloat(Float32)."""
    y = x
    a = d = 1
    b = c = 0
    m = float(_MAXINT_F32)
    while abs(y) <= m:
        f = math.trunc(y)
        y -= f
        a, c = f * a + c, a
        b, d = f * b + d, b
        if max(abs(a), abs(b)) > _MAXINT_F32:
            return c, d
        if b != 0 and a / b == x:
            break
        if y == 0:
            break
        y = 1.0 / y
    return a, b


def _truncbits(x, nb):
    u = struct.unpack("<Q", struct.pack("<d", x))[0]
    u &= (0xFFFFFFFFFFFFFFFF << nb) & 0xFFFFFFFFFFFFFFFF
    return struct.unpack("<d", struct.pack("<Q", u))[0]


def _add12(x, y):
    if abs(y) > abs(x):
        x, y = y, x
    h = x + y
    return h, (x - h) + y


def _nbitslen(ln, offset):
    nb = 0 if ln < 2 else math.ceil(math.log2(max(offset - 1, ln - offset))) + 1
    return min(27, nb)  # cld(precision(Float64), 2)


def julia_range(start, stop, length):
    """collect(range(start, stop, length = length)) for Float64 endpoints, element by element as StepRangeLen's getindex gives them."""
    start, stop, ln = float(start), float(stop), int(length)
    if ln == 1:
        return [start]
    if start == stop:
        return [start] * ln
    # "nice" endpoints: exact rational arithmetic (Base.linspace(T, start_n, stop_n, len, den) evaluates
    # (start_n (len - i) + stop_n (i - 1)) / ((len - 1) den) in twice precision: correctly rounded up to near-ties)
    sn, sd = _rat(start)
    en, ed = _rat(stop)
    if sd != 0 and ed != 0:
        den = sd * ed // math.gcd(sd, ed)
        m = 9007199254740992.0
        if den != 0 and abs(den * start) <= m and abs(den * stop) <= m:
            start_n, stop_n = round(den * start), round(den * stop)
            if start_n / den == start and stop_n / den == stop:
                return [float(Fraction(start_n * (ln - i) + stop_n * (i - 1), (ln - 1) * den)) for i in range(1, ln + 1)]
    d = stop - start
    tmin = -(start / d)
    imin = round(tmin * (ln - 1) + 1)
    if 1 < imin < ln:
        t = (imin - 1) / (ln - 1)
        ref = (1 - t) * start + t * stop
        step = (ref - start) / (imin - 1) if imin - 1 < ln - imin else (stop - ref) / (ln - imin)
    elif imin <= 1:
        imin, ref, step = 1, start, d / (ln - 1)
    else:
        imin, ref, step = ln, stop, d / (ln - 1)
    step_hi = _truncbits(step, _nbitslen(ln, imin))
    x1_hi, x1_lo = _add12((1 - imin) * step_hi, ref)
    x2_hi, x2_lo = _add12((ln - imin) * step_hi, ref)
    a, b = (start - x1_hi) - x1_lo, (stop - x2_hi) - x2_lo
    step_lo = (b - a) / (ln - 1)
    ref_lo = a - (1 - imin) * step_lo
    out = []
    for i in range(1, ln + 1):
        u = i - imin
        shift_hi, shift_lo = u * step_hi, u * step_lo
        x_hi, x_lo = _add12(ref, shift_hi)
        out.append(x_hi + (x_lo + (shift_lo + ref_lo)))
    return out


def sind(x):
    """sind(x) for a finite real x in degrees: Julia reduces mod 360 exactly, then evaluates sin / cos kernels on a double-double
    deg2rad; the result is the correctly rounded sin(x pi / 180) (exact 0, +-0.5, +-1 at the multiples of 30 degrees that have them).
    Restated as a 60-digit evaluation rounded once."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    r = Fraction(x) % 360
    if r in (0, 180):
        return math.copysign(0.0, x) if r == 0 else 0.0
    sign = 1
    if r > 180:
        r, sign = r - 180, -1
    if r > 90:
        r = 180 - r
    exact = {Fraction(30): 0.5, Fraction(90): 1.0}
    if r in exact:
        return sign * exact[r]
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899")
    t = Decimal(r.numerator) / Decimal(r.denominator) * pi / 180
    term, s, n = t, t, 1
    while abs(term) > Decimal(10) ** -58:
        term = -term * t * t / ((2 * n) * (2 * n + 1))
        s += term
        n += 1
    return sign * float(s)
