"""The expression list of the reference's compute/RealTest.scala (rainier-test/.../compute/RealTest.scala:58-204),
written once against a tiny algebra so that the same lambda runs on the RIR authoring DSL (-> oracle interpreter,
-> generated HIP) and on plain IEEE doubles (the "constant folding" leg of the reference test).

RealTest checks, for x in {1, 0, -1, 2, -2, .5, -.5, -Inf, +Inf} (filtered by `defined`):
    constant == reference fn (if given) == Evaluator == compiled IR,            assertWithinEpsilon (rel 1e-3 or NaN == NaN)
    numeric derivative (dx = 1e-5) == symbolic derivative == compiled derivative  where `derivable` and x finite
"""
import math

import numpy as np

POINTS = [1.0, 0.0, -1.0, 2.0, -2.0, 0.5, -0.5, -math.inf, math.inf]


class F:
    """IEEE double with the JVM semantics of the IR ops."""

    def __init__(self, v): self.v = np.float64(v)
    @staticmethod
    def w(o): return o if isinstance(o, F) else F(o)
    def _b(self, o, f):
        with np.errstate(all="ignore"):
            return F(f(self.v, F.w(o).v))
    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return F.w(o) + self
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return F.w(o) - self
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return F.w(o) * self
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return F.w(o) / self
    def __pow__(self, o): return self._b(o, lambda a, b: np.float64(1.0) if b == 0 else np.power(a, b))
    def _u(self, f):
        with np.errstate(all="ignore"):
            return F(f(self.v))
    def exp(self): return self._u(np.exp)
    def log(self): return self._u(np.log)
    def abs(self): return self._u(np.abs)
    def sin(self): return self._u(np.sin)
    def cos(self): return self._u(np.cos)
    def tan(self): return self._u(np.tan)
    def asin(self): return self._u(np.arcsin)
    def acos(self): return self._u(np.arccos)
    def atan(self): return self._u(np.arctan)
    def compare(self, o):  # DCMPL; I2D: NaN -> -1
        a, b = self.v, F.w(o).v
        return F(1.0 if a > b else 0.0 if a == b else -1.0)


class Alg:
    """What the case lambdas may use beyond operators; `g` is a frontend.Graph or None (plain doubles)."""

    def __init__(self, g=None): self.g = g
    def c(self, v): return self.g.const(float(v)) if self.g is not None else F(v)
    def lookup(self, index, table, low=0):
        if self.g is not None:
            return self.g.lookup(index, [t if not isinstance(t, (int, float)) else self.g.const(float(t)) for t in table], low)
        v = index.v
        k = 0 if v != v else int(max(min(v, 2147483647.0), -2147483648.0))  # D2I: truncation, NaN -> 0, saturating
        if not (0 <= k - low < len(table)):
            raise IndexError(k)
        return F.w(table[k - low])
    # Real.eq / Real.gt (compute/Real.scala:83-99): Lookup(Compare(a, b), ..., low = -1)
    def eq(self, a, b, t, f): return self.lookup(a.compare(b), [f, t, f], -1)
    def gt(self, a, b, t, f): return self.lookup(a.compare(b), [f, f, t], -1)
    def max(self, a, b): return self.gt(a, b, a, b)
    def sum(self, xs):
        acc = xs[0]
        for x in xs[1:]:
            acc = acc + x
        return acc
    # Combinatorics.gamma / factorial (core/Combinatorics.scala:10-37): Nemes' log-Gamma approximation
    def log_gamma(self, z):
        v = z + 1.0
        w = v + (1.0 / ((12.0 * v) - (1.0 / (10.0 * v))))
        return (math.log(math.pi * 2) / 2.0) - (v.log() / 2.0) + (v * (w.log() - 1.0)) - z.log()
    def factorial(self, k): return self.log_gamma(k + 1.0)


def _sinh(x): return (x.exp() - (0.0 - x).exp()) / 2.0
def _cosh(x): return (x.exp() + (0.0 - x).exp()) / 2.0
def _normal_logpdf(A, mean, y): return (((A.c(y) - mean) / 1.0) ** 2.0) / -2.0 - 0.5 * math.log(2 * math.pi)  # Continuous.scala:63-67
def _poisson_logpdf(A, lam, v): return lam.log() * v - lam - A.factorial(A.c(v))                                     # Discrete.scala:136-137
def _gamma_std_logpdf(A, shape, y): return (shape - 1.0) * math.log(y) - A.log_gamma(shape) - y                       # Continuous.scala:112-118

_EXPONENTS = [17, -3, 40, -28, 5, 0, -40, 12, 33, -9, 1, -1, 26, -17, 8, -35, 21, 2, -12, 38, -22, 9, 30, -6, 14, -31,
              -2, 36, 4, -19, 24, -38, 11, -8, 19, 28, -14, 6, -25, 31, 3, -33, 16, -5, 22, 39, -11, 7, -27, 34, -16,
              10, -36, 27, -4, 18, -21, 37, 13, -30, 23, -7, 32, -13, 20, -39, 29, -10, 15, -24, 35, -15, 25, -20,
              -18, -23, -26, -29, -32, -34, -37]  # a fixed shuffle of -40..40 (the reference shuffles at random)
assert sorted(_EXPONENTS) == list(range(-40, 41))

finite = lambda x: not math.isinf(x)

# (name, fn(A, x), defined, derivable, reference or None)
CASES = [
    ("plus", lambda A, x: x + 1.0, None, None, None),
    ("exp", lambda A, x: x.exp(), None, None, None),
    ("square", lambda A, x: x * x, None, None, None),
    ("log", lambda A, x: x.abs().log(), None, None, None),
    ("sin", lambda A, x: x.sin(), finite, None, math.sin),
    ("cos", lambda A, x: x.cos(), finite, None, math.cos),
    ("tan", lambda A, x: x.tan(), finite, None, math.tan),
    ("asin", lambda A, x: x.asin(), lambda v: -1 < v < 1, None, math.asin),
    ("acos", lambda A, x: x.acos(), lambda v: -1 < v < 1, None, math.acos),
    ("atan", lambda A, x: x.atan(), None, None, math.atan),
    ("sinh", lambda A, x: _sinh(x), None, None, lambda v: math.copysign(math.inf, v) if math.isinf(v) else math.sinh(v)),
    ("cosh", lambda A, x: _cosh(x), None, None, lambda v: math.inf if math.isinf(v) else math.cosh(v)),
    ("tanh", lambda A, x: _sinh(x) / _cosh(x), finite, None, math.tanh),
    ("tanh at infty", lambda A, x: _sinh(x) / _cosh(x), None, None, None),
    ("cos(x^2)", lambda A, x: (x * x).cos(), finite, None, None),
    ("temp", lambda A, x: (x * 3.0) + (x * 3.0), None, None, None),
    ("abs", lambda A, x: x.abs(), None, None, None),
    ("max(x, 0)", lambda A, x: A.max(x, A.c(0.0)), None, lambda v: v != 0, None),
    ("max(x, x)", lambda A, x: A.max(x, x), None, None, None),
    ("x > 0 ? x^2 : 1", lambda A, x: A.gt(x, A.c(0.0), x * x, A.c(1.0)), None, lambda v: v != 0, None),
    ("x > 0 ? 1 : x + 1", lambda A, x: A.gt(x, A.c(0.0), A.c(1.0), x + 1.0), None, lambda v: v != 0, None),
    ("x > 0 ? x^2 : x + 1", lambda A, x: A.gt(x, A.c(0.0), x * x, x + 1.0), None, lambda v: v != 0, None),
    ("normal", lambda A, x: _normal_logpdf(A, x, 1.0), lambda v: v != math.inf, None, None),
    ("normal sum", lambda A, x: A.sum([_normal_logpdf(A, x, y) for y in (0.0, 1.0, 2.0)]), lambda v: v != math.inf, None, None),
    ("logistic", lambda A, x: ((1.0 / (1.0 + (x * -1.0).exp())) * (1.0 - (1.0 / (1.0 + (x * -1.0).exp())))).log(), None, None, None),
    ("minimal logistic", lambda A, x: 1.0 / (x.exp() + 1.0), None, None, lambda v: 1.0 / (math.exp(v) + 1) if v < 700 else 0.0),
    ("log x^2", lambda A, x: (x ** 2.0).log(), None, lambda v: v != 0, lambda v: math.log(v * v) if v != 0 else -math.inf),
    ("poisson", lambda A, x: A.sum([_poisson_logpdf(A, x.abs() + 1.0, float(y)) for y in range(11)]), None, None, None),
    ("4x^3", lambda A, x: ((((x + x) * x) + (x * x)) * x) + (x * x * x), None, None, lambda v: 4 * v * v * v),
    ("lookup", lambda A, x: A.lookup(x.abs() * 2.0, [0.0, 1.0, 2.0, 3.0, 4.0]), lambda v: abs(v) <= 2 and float(abs(v) * 2).is_integer(),
     lambda v: False, lambda v: abs(v) * 2),
    ("exponent sums", lambda A, x: _exponent_sums(x), None, lambda v: v != 0, None),
    ("cancelling x^2 then distributing", lambda A, x: ((x ** 2.0) * 2.0) / (x ** 2.0) + x, lambda v: v != 0 and finite(v) and float(v).is_integer(), None, None),
    ("pow", lambda A, x: x ** x, lambda v: v >= 0, None, None),
    ("gamma fit", lambda A, x: A.sum([_gamma_std_logpdf(A, x.abs(), y) for y in (1.0, 2.0, 3.0)]), None, None, None),
]


def _exponent_sums(x):
    a = x
    for e in _EXPONENTS:
        a = (a + x ** float(e)) * x
    return a


def within_epsilon(x, y):
    """ComputeTest.assertWithinEpsilon (compute/ComputeTest.scala:6-17)."""
    if abs(x) > 10e-8 or abs(y) > 10e-8:
        if math.isnan(x) and math.isnan(y):
            return True
        if x == y:
            return True
        with np.errstate(all="ignore"):
            rel = abs(np.float64(x - y) / np.float64(x))
        return bool(rel < 0.001)
    return True


def constant(fn, v):
    """the `evalAt` leg: the expression on plain doubles (the reference's constant folding)"""
    try:
        return float(fn(Alg(None), F(v)).v)
    except (IndexError, ZeroDivisionError):
        return math.nan
