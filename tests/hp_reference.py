"""acos, asin, sin, cos to ~100 decimal digits (decimal module: series whose coefficients are exact rationals), for checking that
eo_math.h / eu_math.h return the correctly rounded double.  Test infrastructure; slow (about a millisecond per value)."""
from decimal import Decimal, getcontext

getcontext().prec = 130
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798214808651328230664709384460955058223172535940812848111745")
_EPS = Decimal(10) ** -100


def _asin_small(x):      # |x| <= 1/2: sum (2n)! / (4^n n!^2 (2n+1)) x^(2n+1)
    x2 = x * x
    term = x
    s = x
    n = 0
    while abs(term) > _EPS:
        n += 1
        term = term * x2 * (2 * n - 1) * (2 * n - 1) / ((2 * n) * (2 * n + 1))
        s += term
    return s


def asin(xf):
    x = Decimal(xf)
    if abs(x) <= Decimal("0.5"):
        return _asin_small(x)
    r = PI / 2 - 2 * _asin_small(((1 - abs(x)) / 2).sqrt())
    return r if x > 0 else -r


def acos(xf):
    x = Decimal(xf)
    if abs(x) <= Decimal("0.5"):
        return PI / 2 - _asin_small(x)
    if x > 0:
        return 2 * _asin_small(((1 - x) / 2).sqrt())
    return PI - 2 * _asin_small(((1 + x) / 2).sqrt())


def _reduce(xf):
    x = Decimal(xf)
    return x - (x / (2 * PI)).to_integral_value() * 2 * PI


def sin(xf):
    x = _reduce(xf)
    s = t = x
    x2 = x * x
    n = 1
    while abs(t) > _EPS:
        t = -t * x2 / ((n + 1) * (n + 2))
        n += 2
        s += t
    return s


def cos(xf):
    x = _reduce(xf)
    s = t = Decimal(1)
    x2 = x * x
    n = 0
    while abs(t) > _EPS:
        t = -t * x2 / ((n + 1) * (n + 2))
        n += 2
        s += t
    return s
