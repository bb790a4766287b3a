"""rainier-compute's front half, restated: the `Real` algebra with its Line / LogLine normal
forms, symbolic reverse-mode differentiation, partial evaluation ("inlining") of data, and the Translator that lowers a
Real DAG to the IR the back end receives -- so that the RIR handed to the HIP engine is, node for node, what
`Compiler.compileTargets` would hand to the reference's ASM back end for the same model, not a hand-derived equivalent.

Paths are under /root/reference/rainier-compute/src/main/scala/com/stripe/rainier/ :
    Real / Constant / Scalar / Column / Parameter / Unary / Line / LogLine / Compare / Pow / Lookup   compute/Real.scala:9-339
    RealOps (add, multiply, divide, pow, unary, compare)                                              compute/RealOps.scala:5-99
    LineOps (sum, scale, translate, multiply, log, pow)                                               compute/LineOps.scala:3-97
    LogLineOps (multiply, pow, distribute)                                                            compute/LogLineOps.scala:6-100
    Coefficients (Empty / One / Many; insertion-ordered term list + map)                              compute/Coefficients.scala:5-140
    ConstantOps                                                                                       compute/ConstantOps.scala:5-114
    Gradient.derive                                                                                   compute/Gradient.scala:6-153
    Translator (hash-consing, Line fold, LogLine product tree, x+x / x*x, Lookup sequencing)         compute/Translator.scala:5-188
    Target / TargetGroup / inlinable, PartialEvaluator.inline                                         compute/Target.scala:5-208, PartialEvaluator.scala:3-98
    the IR node set                                                                                   ir/IR.scala:3-51, ir/Ops.scala:3-37
and the serialiser at the bottom is integration/scala/HipCompiler.scala in Python (RIR: include/rainier_hip_rir.h).

Equality is part of the algorithm (hash maps keyed by Real decide which terms merge and which nodes are shared), so it is
restated too: Scalar, Unary, LogLine, Compare, Pow and the Coefficients are Scala case classes (structural equality);
Line, Lookup, Parameter and Column are plain classes (reference equality).  Where the reference iterates a hash set whose
order the JVM does not define (the set of priors, compute/Target.scala:73-75; column sets), insertion order is used.

This module is an AUTHORING / TEST aid on the host (the product boundary is the C ABI; a JVM deployment keeps Rainier's own
front-end and serialises with integration/scala/HipCompiler.scala).  Nothing here runs on the sampling path.
"""
from __future__ import annotations

import math
import struct
import sys
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))

# ir/Ops.scala
ADD, SUB, MUL, DIV, POW, COMPARE = "add", "sub", "mul", "div", "pow", "compare"
EXP, LOG, ABS, NOOP, SIN, COS, TAN, ASIN, ACOS, ATAN = "exp", "log", "abs", "noop", "sin", "cos", "tan", "asin", "acos", "atan"
_COMMUTATIVE = (ADD, MUL)
_RIR_BINARY = {ADD: 2, SUB: 3, MUL: 4, DIV: 5, POW: 6, COMPARE: 7}
_RIR_UNARY = {EXP: 8, LOG: 9, ABS: 10, NOOP: 11, SIN: 12, COS: 13, TAN: 14, ASIN: 15, ACOS: 16, ATAN: 17}


class _Sym:
    """ir/IR.scala:41-51: one global counter for VarDef and Param symbols."""
    n = 0

    @staticmethod
    def fresh() -> int:
        _Sym.n += 1
        return _Sym.n - 1


# ======================================================================================================= Real
class Real:
    """compute/Real.scala:9-44"""
    __slots__ = ()

    def __add__(self, o): return RealOps.add(self, Real.of(o))
    def __radd__(self, o): return RealOps.add(Real.of(o), self)
    def __mul__(self, o): return RealOps.multiply(self, Real.of(o))
    def __rmul__(self, o): return RealOps.multiply(Real.of(o), self)
    def __neg__(self): return self * -1
    def __sub__(self, o): return self + (-Real.of(o))
    def __rsub__(self, o): return Real.of(o) + (-self)
    def __truediv__(self, o): return RealOps.divide(self, Real.of(o))
    def __rtruediv__(self, o): return RealOps.divide(Real.of(o), self)
    def min(self, o): return RealOps.min(self, Real.of(o))
    def max(self, o): return RealOps.max(self, Real.of(o))
    def pow(self, e): return RealOps.pow(self, Real.of(e))
    def exp(self): return RealOps.unary(self, EXP)
    def log(self): return RealOps.unary(self, LOG)
    def sin(self): return RealOps.unary(self, SIN)
    def cos(self): return RealOps.unary(self, COS)
    def tan(self): return RealOps.unary(self, TAN)
    def asin(self): return RealOps.unary(self, ASIN)
    def acos(self): return RealOps.unary(self, ACOS)
    def atan(self): return RealOps.unary(self, ATAN)
    def abs(self): return RealOps.unary(self, ABS)
    def sinh(self): return (self.exp() - (-self).exp()) / 2
    def cosh(self): return (self.exp() + (-self).exp()) / 2
    def tanh(self): return self.sinh() / self.cosh()
    @property
    def logit(self): return -((Real.one / self - 1).log())
    @property
    def logistic(self): return Real.one / (Real.one + (-self).exp())

    # ---- object Real (compute/Real.scala:46-115)
    @staticmethod
    def of(x) -> "Real":
        """ToReal (compute/ToReal.scala:7-24): numbers become Scalars; NaN is an ArithmeticException"""
        if isinstance(x, Real):
            return x
        v = float(x)
        if math.isnan(v):
            raise ArithmeticError("Trying to convert NaN to Real")
        return Scalar(v)

    @staticmethod
    def sum(seq: Iterable) -> "Real":
        acc = Real.zero
        for x in seq:
            acc = acc + x
        return acc

    @staticmethod
    def logSumExp(seq) -> "Real":
        seq = [Real.of(x) for x in seq]
        mx = seq[0]
        for x in seq[1:]:
            mx = mx.max(x)
        return Real.sum([(x - mx).exp() for x in seq]).log() + mx

    @staticmethod
    def parameter(fn=None) -> "Parameter":
        x = Parameter(Prior(Real.zero))
        if fn is not None:
            x.prior = Prior(Real.of(fn(x)))
        return x

    @staticmethod
    def parameters(size: int, fn) -> List["Parameter"]:
        vec = [Parameter(Prior(Real.zero)) for _ in range(size)]
        prior = Prior(Real.of(fn(vec)))
        for x in vec:
            x.prior = prior
        return vec

    @staticmethod
    def doubles(seq) -> "Column": return Column(np.asarray(seq, dtype=np.float64))
    @staticmethod
    def longs(seq) -> "Column": return Column(np.asarray(seq, dtype=np.float64))

    @staticmethod
    def _lookupCompare(left, right, gt, eq, lt):
        return Lookup.apply(RealOps.compare(Real.of(left), Real.of(right)), [Real.of(lt), Real.of(eq), Real.of(gt)], -1)
    @staticmethod
    def eq(l, r, t, f): return Real._lookupCompare(l, r, f, t, f)
    @staticmethod
    def lt(l, r, t, f): return Real._lookupCompare(l, r, f, f, t)
    @staticmethod
    def gt(l, r, t, f): return Real._lookupCompare(l, r, t, f, f)
    @staticmethod
    def lte(l, r, t, f): return Real._lookupCompare(l, r, f, t, t)
    @staticmethod
    def gte(l, r, t, f): return Real._lookupCompare(l, r, t, t, f)


class Constant(Real):
    """compute/Real.scala:117-148"""
    __slots__ = ()
    @property
    def isZero(self): return self.lower == 0.0 and self.upper == 0.0
    @property
    def isOne(self): return self.lower == 1.0 and self.upper == 1.0
    @property
    def isTwo(self): return self.lower == 2.0 and self.upper == 2.0
    @property
    def isPosInfinity(self): return self.lower == math.inf and self.upper == math.inf
    @property
    def isNegInfinity(self): return self.lower == -math.inf and self.upper == -math.inf
    @property
    def isPositive(self): return self.lower >= 0.0


class Scalar(Constant):
    """final private case class Scalar(value: Double) (compute/Real.scala:150-162): equal iff the doubles are =="""
    __slots__ = ("value",)

    def __init__(self, value: float):
        self.value = float(value)
    @property
    def lower(self): return self.value
    @property
    def upper(self): return self.value
    def getDouble(self): return self.value
    def map(self, fn): return Scalar(fn(np.float64(self.value)))
    def mapWith(self, other: Constant, fn):
        if isinstance(other, Scalar):
            with np.errstate(all="ignore"):
                return Scalar(float(fn(np.float64(self.value), np.float64(other.value))))
        with np.errstate(all="ignore"):
            return Column(fn(np.float64(self.value), other.values))
    def __eq__(self, o): return isinstance(o, Scalar) and self.value == o.value
    def __hash__(self): return hash(self.value)
    def __repr__(self): return "Scalar(%r)" % self.value


class Column(Constant):
    """final class Column(val values: Array[Double]) extends Constant (compute/Real.scala:164-187): one value per observation
    row; reference equality; owns an ir.Param (its slot in the DataFunction input layout)."""
    __slots__ = ("values", "sym", "lower", "upper")

    def __init__(self, values):
        self.values = np.ascontiguousarray(values, dtype=np.float64)
        self.sym = _Sym.fresh()
        self.lower = float(self.values.min()) if self.values.size else math.inf
        self.upper = float(self.values.max()) if self.values.size else -math.inf
    def getDouble(self): raise RuntimeError("Not a scalar")
    def map(self, fn):
        with np.errstate(all="ignore"):
            return Column(fn(self.values))
    def mapWith(self, other: Constant, fn):
        with np.errstate(all="ignore"):
            if isinstance(other, Scalar):
                return Column(fn(self.values, np.float64(other.value)))
            return Column(fn(self.values, other.values))
    @property
    def maybeScalar(self) -> Optional[float]:
        return self.lower if self.lower == self.upper else None
    __hash__ = object.__hash__
    def __eq__(self, o): return self is o
    def __repr__(self): return "Column(n=%d)" % self.values.size


class NonConstant(Real):
    __slots__ = ()


class Prior:
    __slots__ = ("density",)
    def __init__(self, density: Real): self.density = density


class Parameter(NonConstant):
    """final class Parameter(var prior: Prior) (compute/Real.scala:191-196); ordered by its ir.Param symbol id"""
    __slots__ = ("prior", "sym")
    def __init__(self, prior: Prior):
        self.prior = prior
        self.sym = _Sym.fresh()
    __hash__ = object.__hash__
    def __eq__(self, o): return self is o
    def __repr__(self): return "Parameter(%d)" % self.sym


class Unary(NonConstant):
    """case class Unary(original: NonConstant, op: ir.UnaryOp) (compute/Real.scala:198-213)"""
    __slots__ = ("original", "op", "_h")
    def __init__(self, original: NonConstant, op: str):
        self.original, self.op = original, op
        self._h = hash(("U", op, hash(original)))
    def __hash__(self): return self._h
    def __eq__(self, o):
        return self is o or (isinstance(o, Unary) and self._h == o._h and self.op == o.op and self.original == o.original)
    def __repr__(self): return "%s(%r)" % (self.op, self.original)


class Line(NonConstant):
    """ax + b with constant a, b (compute/Real.scala:215-241).  Deliberately NOT a case class: reference equality."""
    __slots__ = ("ax", "b")
    def __init__(self, ax: "Coefficients", b: Constant):
        assert not ax.isEmpty                                     # require(!ax.isEmpty)
        self.ax, self.b = ax, b
    __hash__ = object.__hash__
    def __eq__(self, o): return self is o
    def __repr__(self): return "Line(%r, %r)" % (self.ax.toList(), self.b)


class LogLine(NonConstant):
    """x^a * y^b * ... with constant exponents (compute/Real.scala:243-272); a case class over its Coefficients"""
    __slots__ = ("ax", "_h")
    def __init__(self, ax: "Coefficients"):
        assert not ax.isEmpty
        self.ax = ax
        self._h = hash(("LL", hash(ax)))
    @staticmethod
    def apply(nc: NonConstant) -> "LogLine":
        return nc if isinstance(nc, LogLine) else LogLine(Coefficients.one_term(nc))
    def __hash__(self): return self._h
    def __eq__(self, o): return self is o or (isinstance(o, LogLine) and self._h == o._h and self.ax == o.ax)
    def __repr__(self): return "LogLine(%r)" % (self.ax.toList(),)


class Compare(NonConstant):
    """case class Compare(left, right): 0 if equal, 1 if left > right, -1 if left < right (compute/Real.scala:274-282)"""
    __slots__ = ("left", "right", "_h")
    def __init__(self, left: Real, right: Real):
        self.left, self.right = left, right
        self._h = hash(("C", hash(left), hash(right)))
    def __hash__(self): return self._h
    def __eq__(self, o):
        return self is o or (isinstance(o, Compare) and self._h == o._h and self.left == o.left and self.right == o.right)


class Pow(NonConstant):
    """case class Pow(base: Real, exponent: NonConstant) (compute/Real.scala:284-287)"""
    __slots__ = ("base", "exponent", "_h")
    def __init__(self, base: Real, exponent: NonConstant):
        self.base, self.exponent = base, exponent
        self._h = hash(("P", hash(base), hash(exponent)))
    def __hash__(self): return self._h
    def __eq__(self, o):
        return self is o or (isinstance(o, Pow) and self._h == o._h and self.base == o.base and self.exponent == o.exponent)


class Lookup(NonConstant):
    """final class Lookup(index, table, low): the (index-low)'th element of table (compute/Real.scala:289-339)"""
    __slots__ = ("index", "table", "low")
    def __init__(self, index: Real, table: Sequence[Real], low: int):
        self.index, self.table, self.low = index, list(table), int(low)
    __hash__ = object.__hash__
    def __eq__(self, o): return self is o

    @staticmethod
    def apply(index: Real, table: Sequence[Real], low: int = 0) -> Real:
        def pick(v: float) -> Real:
            if float(v).is_integer():
                return table[int(v) - low]
            raise ArithmeticError("Cannot lookup a non-integral number")
        if isinstance(index, Scalar):
            return pick(index.value)
        if isinstance(index, Column):
            ms = index.maybeScalar
            if ms is not None:
                return pick(ms)
            if all(isinstance(t, Scalar) for t in table):
                scalars = np.array([t.value for t in table])
                if not np.all(np.floor(index.values) == index.values):
                    raise ArithmeticError("Cannot lookup a non-integral number")
                return Column(scalars[index.values.astype(np.int64) - low])
            return Lookup(index, table, low)
        return Lookup(index, table, low)


Real.zero = Scalar(0.0)
Real.one = Scalar(1.0)
Real.two = Scalar(2.0)
Real.negOne = Scalar(-1.0)
Real.Pi = Scalar(math.pi)
Real.infinity = Scalar(math.inf)
Real.negInfinity = Scalar(-math.inf)
_ZERO, _ONE, _TWO, _NEG_TWO, _PI, _INF, _NINF = Real.zero, Real.one, Real.two, Scalar(-2.0), Real.Pi, Real.infinity, Real.negInfinity


# ======================================================================================================= Coefficients
class Coefficients:
    """compute/Coefficients.scala:5-140.  kind 0 = Empty, 1 = One(term, coefficient), 2 = Many(toMap, terms): the term LIST
    keeps insertion order (new terms are prepended) and is what every traversal uses; the map is for membership."""
    __slots__ = ("kind", "term", "coefficient", "map", "terms", "_h")

    def __init__(self, kind, term=None, coefficient=None, map=None, terms=None):
        self.kind, self.term, self.coefficient, self.map, self.terms = kind, term, coefficient, map, terms
        if kind == 0: self._h = 0
        elif kind == 1: self._h = hash(("One", hash(term), hash(coefficient)))
        else: self._h = hash(("Many", tuple(hash(t) for t in terms)))

    # -- constructors (object Coefficients)
    @staticmethod
    def one_term(term: NonConstant) -> "Coefficients": return Coefficients.pair(term, _ONE)
    @staticmethod
    def pair(term: NonConstant, c: Constant) -> "Coefficients":
        return Coefficients.Empty if c.isZero else Coefficients(1, term, c)
    @staticmethod
    def seq(pairs: Sequence[Tuple[NonConstant, Constant]]) -> "Coefficients":
        filtered = [(x, a) for x, a in pairs if not a.isZero]
        if not filtered: return Coefficients.Empty
        if len(filtered) == 1: return Coefficients.pair(*filtered[0])
        return Coefficients(2, map=dict(filtered), terms=[x for x, _ in filtered])

    # -- queries
    @property
    def isEmpty(self): return self.kind == 0
    @property
    def size(self): return 0 if self.kind == 0 else (1 if self.kind == 1 else len(self.map))
    def toList(self) -> List[Tuple[NonConstant, Constant]]:
        if self.kind == 0: return []
        if self.kind == 1: return [(self.term, self.coefficient)]
        return [(x, self.map[x]) for x in self.terms]

    def withComplements(self):
        if self.kind == 0: return []
        if self.kind == 1: return [(self.term, self.coefficient, Coefficients.Empty)]
        acc, a, b = [], [], list(self.terms)
        while b:
            head, tail = b[0], b[1:]
            ct = (tail + a) if len(a) > len(tail) else (a + tail)
            if len(ct) == 1:
                comp = Coefficients(1, ct[0], self.map[ct[0]])
            else:
                m = dict(self.map); del m[head]
                comp = Coefficients(2, map=m, terms=ct)
            acc.insert(0, (head, self.map[head], comp))
            a, b = [head] + a, tail
        return acc

    def mapCoefficients(self, fn) -> "Coefficients":
        if self.kind == 0: return self
        if self.kind == 1: return Coefficients(1, self.term, fn(self.coefficient))
        return Coefficients(2, map={x: fn(a) for x, a in self.map.items()}, terms=list(self.terms))

    def merge(self, other: "Coefficients") -> "Coefficients":
        if self.kind == 0: return other
        if self.kind == 1: return other.plus(self.term, self.coefficient)
        if other.size > self.size:
            return other.merge(self)
        acc = self
        for x, a in other.toList():
            acc = acc.plus(x, a)
        return acc

    def plus(self, term: NonConstant, coefficient: Constant) -> "Coefficients":
        if self.kind == 0:
            return Coefficients.pair(term, coefficient)
        if self.kind == 1:
            if term == self.term:
                nc = ConstantOps.add(self.coefficient, coefficient)
                return Coefficients.Empty if nc.isZero else Coefficients(1, self.term, nc)
            return Coefficients.seq([(term, coefficient)] + self.toList())
        if term in self.map:
            nc = ConstantOps.add(coefficient, self.map[term])
            if nc.isZero:
                m = dict(self.map); del m[term]
                nt = [t for t in self.terms if not (t == term)]
                if len(nt) == 1:
                    return Coefficients(1, nt[0], next(iter(m.values())))
                return Coefficients(2, map=m, terms=nt)
            m = dict(self.map)
            m[term] = nc
            return Coefficients(2, map=m, terms=self.terms)
        m = dict(self.map); m[term] = coefficient
        return Coefficients(2, map=m, terms=[term] + self.terms)

    def __hash__(self): return self._h
    def __eq__(self, o):
        if self is o: return True
        if not isinstance(o, Coefficients) or self.kind != o.kind or self._h != o._h: return False
        if self.kind == 0: return True
        if self.kind == 1: return self.term == o.term and self.coefficient == o.coefficient
        return self.terms == o.terms and self.map == o.map


Coefficients.Empty = Coefficients(0)


# ======================================================================================================= ConstantOps
class ConstantOps:
    """compute/ConstantOps.scala:5-114"""

    @staticmethod
    def unary(c: Constant, op: str) -> Constant:
        def fail(msg): raise ArithmeticError(msg)
        if c.isPosInfinity:
            if op in (EXP, LOG, ABS, NOOP): return _INF
            if op == ATAN: return ConstantOps.divide(_PI, _TWO)
            fail("No limit for '%s' at positive infinity" % op)
        if c.isNegInfinity:
            if op == EXP: return _ZERO
            if op == ABS: return _INF
            if op == ATAN: return ConstantOps.divide(_PI, _NEG_TWO)
            if op == NOOP: return c
            fail("Cannot take the log of a negative number" if op == LOG else "No limit for '%s' at negative infinity" % op)
        if c.isZero:
            if op in (EXP, COS): return _ONE
            if op == LOG: return _NINF
            if op == ACOS: return ConstantOps.divide(_PI, _TWO)
            if op == NOOP: return c
            return _ZERO
        if op == LOG and not c.isPositive:
            fail("Cannot take the log of a negative number")
        if op == NOOP:
            return c
        fn = {EXP: np.exp, LOG: np.log, ABS: np.abs, SIN: np.sin, COS: np.cos, TAN: np.tan, ASIN: np.arcsin, ACOS: np.arccos,
              ATAN: np.arctan}[op]
        with np.errstate(all="ignore"):
            return c.map(lambda v: fn(v)) if isinstance(c, Column) else Scalar(float(fn(np.float64(c.value))))

    @staticmethod
    def add(l: Constant, r: Constant) -> Constant:
        if (l.isNegInfinity and r.isPosInfinity) or (l.isPosInfinity and r.isNegInfinity):
            raise ArithmeticError("Cannot add +inf and -inf")
        return l.mapWith(r, lambda a, b: a + b)
    @staticmethod
    def multiply(l: Constant, r: Constant) -> Constant:
        if ((l.isPosInfinity or l.isNegInfinity) and r.isZero) or (l.isZero and (r.isPosInfinity or r.isNegInfinity)):
            raise ArithmeticError("Cannot multiply inf by zero")
        return l.mapWith(r, lambda a, b: a * b)
    @staticmethod
    def divide(l: Constant, r: Constant) -> Constant:
        if l.isZero and r.isZero:
            raise ArithmeticError("Cannot divide zero by zero")
        return l.mapWith(r, lambda a, b: a / b)
    @staticmethod
    def pow(l: Constant, r: Constant) -> Constant:
        return l.mapWith(r, _java_pow)
    @staticmethod
    def compare(l: Constant, r: Constant) -> Constant:
        def cmp(a, b):
            a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
            eq = (a == b)
            lt = (a < b) | (a == -np.inf) | (b == np.inf)
            return np.where(eq, 0.0, np.where(lt, -1.0, 1.0))
        return l.mapWith(r, cmp)


def _java_pow(x, y):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    with np.errstate(all="ignore"):
        v = np.power(x, y)
        v = np.where(np.isnan(y), np.nan, v)
        v = np.where(np.isinf(y) & (np.abs(x) == 1.0), np.nan, v)
        return np.where(y == 0.0, 1.0, v)



# ======================================================================================================= Bounds
class Bounds:
    """compute/Bounds.scala:5-141: interval arithmetic over the Real DAG.  It decides whether a density is wrapped in a
    Real.gte(...) guard (Bounds.positive / zeroToOne), so it is part of the lowering, not a diagnostic."""

    @staticmethod
    def or_(seq): return (min(b[0] for b in seq), max(b[1] for b in seq))
    @staticmethod
    def sum(seq):
        lo = hi = 0.0
        for b in seq:
            lo += b[0]; hi += b[1]
        return (lo, hi)
    @staticmethod
    def _mul(l, r):
        if math.isinf(l) and r == 0.0: return l
        if l == 0.0 and math.isinf(r): return r
        return l * r
    @staticmethod
    def multiply(l, r):
        o = [Bounds._mul(l[0], r[0]), Bounds._mul(l[0], r[1]), Bounds._mul(l[1], r[0]), Bounds._mul(l[1], r[1])]
        return (_nanmin(o), _nanmax(o))
    @staticmethod
    def pow(x, y):
        if y[0] >= 0.0: return Bounds._positivePow(x, y)
        if y[1] <= 0.0: return Bounds._negativePow(x, y)
        return Bounds.or_([Bounds._negativePow(x, (y[0], 0.0)), Bounds._positivePow(x, (0.0, y[1]))])
    @staticmethod
    def _positivePow(x, y):
        if x[0] >= 0.0: return Bounds._pp(x, y)
        if x[1] <= 0.0: return Bounds._np(x, y)
        return Bounds.or_([Bounds._np((x[0], 0.0), y), Bounds._pp((0.0, x[1]), y)])
    @staticmethod
    def _negativePow(x, y): return Bounds.reciprocal(Bounds._positivePow(x, (y[0] * -1, y[1] * -1)))
    @staticmethod
    def _pp(x, y):
        o = [_jpow(x[0], y[0]), _jpow(x[0], y[1]), _jpow(x[1], y[0]), _jpow(x[1], y[1])]
        return (_nanmin(o), _nanmax(o))
    @staticmethod
    def _np(x, y):
        if y[0] == y[1] and float(y[0]).is_integer() and abs(y[0]) < 2 ** 31:
            o = [_jpow(x[0], y[0]), _jpow(x[1], y[0])]
            return (_nanmin(o), _nanmax(o))
        return (-math.inf, math.inf)
    @staticmethod
    def reciprocal(x):
        if x[0] <= 0.0 and x[1] >= 0.0: return (-math.inf, math.inf)
        return (1.0 / x[1], 1.0 / x[0])
    @staticmethod
    def abs(x):
        if x[0] <= 0.0 and x[1] >= 0.0: return (0.0, max(abs(x[0]), x[1]))
        o = [abs(x[0]), abs(x[1])]
        return (min(o), max(o))

    # -- the guards (Bounds.scala:106-134)
    @staticmethod
    def test(value: "Real", fn) -> bool:
        lo, hi = bounds(Real.of(value))
        return bool(fn(lo)) and bool(fn(hi))
    @staticmethod
    def positive(value: "Real", calc) -> "Real":
        if Bounds.test(value, lambda v: v >= 0.0): return calc()
        return Real.gte(value, Real.zero, calc(), Real.negInfinity)
    @staticmethod
    def zeroToOne(value: "Real", calc) -> "Real":
        if Bounds.test(value, lambda v: v >= 0.0 and v <= 1.0): return calc()
        return Real.gte(value, Real.zero, Real.lte(value, Real.one, calc(), Real.negInfinity), Real.negInfinity)


def _nanmin(o):  # Scala's List[Double].min: NaN-propagating is not specified; math.min semantics (NaN wins)
    return math.nan if any(v != v for v in o) else min(o)
def _nanmax(o):
    return math.nan if any(v != v for v in o) else max(o)
def _jpow(x, y): return float(_java_pow(x, y))
def _jlog(x):
    with np.errstate(all="ignore"):
        return float(np.log(np.float64(x)))
def _jexp(x):
    with np.errstate(all="ignore"):
        return float(np.exp(np.float64(x)))

_BOUNDS: Dict[int, tuple] = {}

def bounds(r: "Real") -> Tuple[float, float]:
    """`val bounds` of every Real node (compute/Real.scala): computed on demand and cached per object"""
    if isinstance(r, Constant):
        return (r.lower, r.upper)
    k = id(r)
    hit = _BOUNDS.get(k)
    if hit is not None and hit[0] is r:
        return hit[1]
    if isinstance(r, Parameter): b = (-math.inf, math.inf)
    elif isinstance(r, Unary):
        o = bounds(r.original)
        if r.op == NOOP: b = o
        elif r.op == ABS: b = Bounds.abs(o)
        elif r.op == EXP: b = (_jexp(o[0]), _jexp(o[1]))
        elif r.op == LOG: b = (_jlog(o[0]), _jlog(o[1]))
        elif r.op in (SIN, COS): b = (-1.0, 1.0)
        elif r.op == TAN: b = (-math.inf, math.inf)
        else: b = (0.0, math.pi / 2.0)                        # asin / acos / atan: as the reference has it ("todo: narrow")
    elif isinstance(r, Line):
        b = Bounds.sum([(r.b.lower, r.b.upper)] + [Bounds.multiply(bounds(x), (a.lower, a.upper)) for x, a in r.ax.toList()])
    elif isinstance(r, LogLine):
        bs = [Bounds.pow(bounds(x), (a.lower, a.upper)) for x, a in r.ax.toList()]
        b = bs[0]
        for nb in bs[1:]:
            b = Bounds.multiply(b, nb)
    elif isinstance(r, Compare): b = (-1.0, 1.0)
    elif isinstance(r, Pow): b = Bounds.pow(bounds(r.base), bounds(r.exponent))
    elif isinstance(r, Lookup): b = Bounds.or_([bounds(t) for t in r.table])
    else: raise TypeError(type(r))
    _BOUNDS[k] = (r, b)
    return b

# ======================================================================================================= RealOps
class RealOps:
    """compute/RealOps.scala:5-99"""

    @staticmethod
    def unary(original: Real, op: str) -> Real:
        if isinstance(original, Constant):
            return ConstantOps.unary(original, op)
        nc = original
        if op == EXP and isinstance(nc, Unary) and nc.op == LOG: return nc.original
        if op == ABS and isinstance(nc, Unary) and nc.op in (ABS, EXP): return nc
        if op == LOG and isinstance(nc, Unary) and nc.op == EXP: return nc.original
        if op == LOG and isinstance(nc, Line):
            r = LineOps.log(nc)
            if r is not None: return r
        # (LogOp, l: LogLine) => LogLineOps.log(l) is a placeholder that always answers None
        return Unary(nc, op)

    @staticmethod
    def add(left: Real, right: Real) -> Real:
        lc, rc = isinstance(left, Constant), isinstance(right, Constant)
        if lc and rc: return ConstantOps.add(left, right)
        if left == _INF: return left
        if right == _INF: return right
        if left == _NINF: return left
        if right == _NINF: return right
        if right == _ZERO: return left
        if left == _ZERO: return right
        if lc: return LineOps.translate(right, left)
        if rc: return LineOps.translate(left, right)
        return LineOps.sum(left, right)

    @staticmethod
    def multiply(left: Real, right: Real) -> Real:
        lc, rc = isinstance(left, Constant), isinstance(right, Constant)
        if lc and rc: return ConstantOps.multiply(left, right)
        if left == _INF: return Real.gt(right, 0, _INF, _NINF)
        if right == _INF: return Real.gt(left, 0, _INF, _NINF)
        if left == _NINF: return Real.gt(right, Real.zero, _NINF, _INF)
        if right == _NINF: return Real.gt(left, Real.zero, _NINF, _INF)
        if right == _ZERO or left == _ZERO: return Real.zero
        if right == _ONE: return left
        if left == _ONE: return right
        if lc: return LineOps.scale(right, left)
        if rc: return LineOps.scale(left, right)
        return LogLineOps.multiply(LogLine.apply(left), LogLine.apply(right))

    @staticmethod
    def divide(left: Real, right: Real) -> Real:
        if isinstance(left, Constant) and isinstance(right, Constant): return ConstantOps.divide(left, right)
        if right == _ZERO: return left * _INF
        return left * right.pow(-1)

    @staticmethod
    def min(left: Real, right: Real) -> Real: return Real.lt(left, right, left, right)
    @staticmethod
    def max(left: Real, right: Real) -> Real: return Real.gt(left, right, left, right)

    @staticmethod
    def pow(original: Real, exponent: Real) -> Real:
        if not isinstance(exponent, Constant):
            return Pow(original, exponent)
        if isinstance(original, Constant): return ConstantOps.pow(original, exponent)
        if exponent == _INF: return _INF
        if exponent == _NINF: return _ZERO
        if exponent == _ZERO: return _ONE
        if exponent == _ONE: return original
        if isinstance(original, Line):
            r = LineOps.pow(original, exponent)
            if r is not None: return r
        return LogLineOps.pow(LogLine.apply(original), exponent)

    @staticmethod
    def compare(left: Real, right: Real) -> Real:
        if isinstance(left, Constant) and isinstance(right, Constant): return ConstantOps.compare(left, right)
        if left == _INF: return _ONE
        if right == _INF: return Real.negOne
        if left == _NINF: return Real.negOne
        if right == _NINF: return _ONE
        return Compare(left, right)


class LineOps:
    """compute/LineOps.scala:3-97"""

    @staticmethod
    def axb(nc: NonConstant) -> Tuple[Coefficients, Constant]:
        if isinstance(nc, Line): return nc.ax, nc.b
        if isinstance(nc, LogLine):
            d = LogLineOps.distribute(nc)
            return d if d is not None else (Coefficients.one_term(nc), _ZERO)
        return Coefficients.one_term(nc), _ZERO

    @staticmethod
    def sum(left: NonConstant, right: NonConstant) -> Real:
        lax, lb = LineOps.axb(left)
        rax, rb = LineOps.axb(right)
        merged = lax.merge(rax)
        if merged.isEmpty:
            return ConstantOps.add(lb, rb)
        return LineOps.simplify(merged, ConstantOps.add(lb, rb))

    @staticmethod
    def scale(nc: NonConstant, v: Constant) -> Real:
        ax, b = LineOps.axb(nc)
        return LineOps.simplify(ax.mapCoefficients(lambda a: ConstantOps.multiply(a, v)), ConstantOps.multiply(b, v))

    @staticmethod
    def translate(nc: NonConstant, v: Constant) -> Real:
        ax, b = LineOps.axb(nc)
        return LineOps.simplify(ax, ConstantOps.add(b, v))

    @staticmethod
    def multiply(left: Line, right: Line) -> Line:
        allLeft = [(Real.one, left.b)] + left.ax.toList()
        allRight = [(Real.one, right.b)] + right.ax.toList()
        nAx, nB = Coefficients.Empty, _ZERO
        for x, a in allLeft:
            for y, c in allRight:
                xy, ac = x * y, ConstantOps.multiply(a, c)
                if isinstance(xy, NonConstant):
                    nAx = nAx.merge(Coefficients.pair(xy, ac))
                else:
                    nB = ConstantOps.add(nB, ConstantOps.multiply(xy, ac))
        return Line(nAx, nB)

    @staticmethod
    def log(line: Line) -> Optional[Real]:
        if line.ax.kind == 1 and line.ax.coefficient.isPositive and line.b.isZero:
            return line.ax.term.log() + ConstantOps.unary(line.ax.coefficient, LOG)
        return None

    @staticmethod
    def pow(line: Line, exponent: Constant) -> Optional[Real]:
        if line.ax.kind == 1 and line.b.isZero:
            return line.ax.term.pow(exponent) * RealOps.pow(line.ax.coefficient, exponent)
        return None

    @staticmethod
    def simplify(ax: Coefficients, b: Constant) -> Real:
        if ax.kind == 0: return b
        if ax.kind == 1 and ax.coefficient.isOne and b.isZero: return ax.term
        return Line(ax, b)


class LogLineOps:
    """compute/LogLineOps.scala:6-100"""
    DistributeToMaxTerms = 20

    @staticmethod
    def multiply(left: LogLine, right: LogLine) -> Real:
        merged = left.ax.merge(right.ax)
        return Real.one if merged.isEmpty else LogLine(merged)

    @staticmethod
    def pow(line: LogLine, v: Constant) -> LogLine:
        return LogLine(line.ax.mapCoefficients(lambda a: ConstantOps.multiply(a, v)))

    @staticmethod
    def distribute(line: LogLine) -> Optional[Tuple[Coefficients, Constant]]:
        MAX = LogLineOps.DistributeToMaxTerms
        def nTerms(l: Line): return l.ax.size if l.b.isZero else l.ax.size + 1
        def nTerms2(l: Line):
            n = nTerms(l)
            return (n * (n + 1)) // 2
        factors: List[Tuple[NonConstant, Constant]] = []
        terms: Optional[Line] = None
        for x, c in line.ax.toList():
            if isinstance(x, Line) and terms is None and c.isOne and nTerms(x) < MAX:
                terms = x
            elif isinstance(x, Line) and terms is not None and c.isOne and nTerms(terms) * nTerms(x) < MAX:
                terms = LineOps.multiply(terms, x)
            elif isinstance(x, Line) and terms is None and c.isTwo and nTerms2(x) < MAX:
                terms = LineOps.multiply(x, x)
            elif isinstance(x, Line) and terms is not None and c.isTwo and nTerms(terms) * nTerms2(x) < MAX:
                terms = LineOps.multiply(terms, LineOps.multiply(x, x))
            else:
                factors.insert(0, (x, c))
        if terms is None:
            return None
        l = terms
        if not factors:
            return l.ax, l.b
        ll = LogLine(Coefficients.seq(factors))
        nAx, nB = Coefficients.pair(ll, l.b), _ZERO
        for x, a in l.ax.toList():
            m = LogLineOps.multiply(ll, LogLine.apply(x))
            if isinstance(m, Constant):
                nB = ConstantOps.add(nB, ConstantOps.multiply(m, a))
            else:
                nAx = nAx.merge(Coefficients.pair(m, a))
        return nAx, nB


# ======================================================================================================= Gradient
class Gradient:
    """compute/Gradient.scala:6-153: reverse-mode differentiation ON THE Real DAG (the result is again a list of Reals)"""

    @staticmethod
    def derive(parameters: Sequence[Parameter], output: Real) -> List[Real]:
        diffs: Dict[Real, "_CompoundDiff"] = {}

        def diff(r: Real) -> "_CompoundDiff":
            d = diffs.get(r)
            if d is None:
                d = diffs[r] = _CompoundDiff()
            return d

        diff(output).register(_ConstDiff())
        visited = set()

        def visit(real: Real):
            # depth-first recursion like the reference: the order of the register() calls is part of the result
            if real in visited:
                return
            visited.add(real)
            if isinstance(real, (Parameter, Constant)):
                return
            if isinstance(real, Pow):
                diff(real.base).register(_PowDiff(real, diff(real), False))
                diff(real.exponent).register(_PowDiff(real, diff(real), True))
                visit(real.base); visit(real.exponent)
            elif isinstance(real, Unary):
                diff(real.original).register(_UnaryDiff(real, diff(real)))
                visit(real.original)
            elif isinstance(real, Line):
                for x, a in real.ax.toList():
                    diff(x).register(_ProductDiff(a, diff(real)))
                    visit(x)
            elif isinstance(real, LogLine):
                for x, a, c in real.ax.withComplements():
                    diff(x).register(_LogLineDiff(diff(real), x, a, c))
                    visit(x)
            elif isinstance(real, Lookup):
                for i, x in enumerate(real.table):
                    diff(x).register(_LookupDiff(real, diff(real), i + real.low))
                    visit(x)
                visit(real.index)
            elif isinstance(real, Compare):
                visit(real.left); visit(real.right)

        visit(output)
        return [diff(v).toReal() for v in parameters]


class _ConstDiff:
    def toReal(self): return Real.one

class _CompoundDiff:
    __slots__ = ("parts", "_real")
    def __init__(self): self.parts, self._real = [], None
    def register(self, part): self.parts.insert(0, part)
    def toReal(self) -> Real:
        if self._real is None:
            self._real = self.parts[0].toReal() if len(self.parts) == 1 else Real.sum([p.toReal() for p in self.parts])
        return self._real

class _ProductDiff:
    def __init__(self, other: Constant, gradient): self.other, self.gradient = other, gradient
    def toReal(self): return self.gradient.toReal() * self.other

class _UnaryDiff:
    def __init__(self, child: Unary, gradient): self.child, self.gradient = child, gradient
    def toReal(self):
        c, g = self.child, self.gradient.toReal()
        op, x = c.op, c.original
        if op == LOG: return g * (Real.one / x)
        if op == EXP: return g * c
        if op == ABS: return Real.eq(x, Real.zero, Real.zero, g * x / c)
        if op == NOOP: return g
        if op == SIN: return g * x.cos()
        if op == COS: return g * (Real.zero - x.sin())
        if op == TAN: return g / x.cos().pow(2)
        if op == ASIN: return g / (Real.one - x.pow(2)).pow(0.5)
        if op == ACOS: return -g / (Real.one - x.pow(2)).pow(0.5)
        if op == ATAN: return g / (Real.one + x.pow(2))
        raise ValueError(op)

class _PowDiff:
    def __init__(self, child: Pow, gradient, isExponent: bool): self.child, self.gradient, self.isExponent = child, gradient, isExponent
    def toReal(self):
        c, g = self.child, self.gradient.toReal()
        if self.isExponent:
            return g * c * Real.eq(c.base, Real.zero, Real.one, c.base).log()
        return g * c.exponent * c.base.pow(c.exponent - 1)

class _LogLineDiff:
    def __init__(self, gradient, term: NonConstant, exponent: Constant, complement: Coefficients):
        self.gradient, self.term, self.exponent, self.complement = gradient, term, exponent, complement
    def toReal(self):
        other = Real.one if self.complement.isEmpty else LogLine(self.complement)
        return self.gradient.toReal() * self.exponent * self.term.pow(self.exponent - Real.one) * other

class _LookupDiff:
    def __init__(self, child: Lookup, gradient, index: int): self.child, self.gradient, self.index = child, gradient, index
    def toReal(self): return Real.eq(self.child.index, self.index, self.gradient.toReal(), Real.zero)


# ======================================================================================================= IR + Translator
class Const:
    __slots__ = ("value",)
    def __init__(self, value: float): self.value = float(value)
class Param:
    """ir.Param of a Parameter or a Column: identified by the owner's symbol"""
    __slots__ = ("sym",)
    def __init__(self, sym: int): self.sym = sym
class VarRef:
    __slots__ = ("sym",)
    def __init__(self, sym: int): self.sym = sym
class VarDef:
    __slots__ = ("sym", "rhs")
    def __init__(self, rhs, sym: int = None):
        self.sym = _Sym.fresh() if sym is None else sym
        self.rhs = rhs
class BinaryIR:
    __slots__ = ("left", "right", "op")
    def __init__(self, left, right, op): self.left, self.right, self.op = left, right, op
class UnaryIR:
    __slots__ = ("original", "op")
    def __init__(self, original, op): self.original, self.op = original, op
class LookupIR:
    __slots__ = ("index", "table", "low")
    def __init__(self, index, table, low): self.index, self.table, self.low = index, table, low
class SeqIR:
    __slots__ = ("first", "second")
    def __init__(self, first: VarDef, second: VarDef): self.first, self.second = first, second

    @staticmethod
    def of(defs: List[VarDef]) -> VarDef:                                      # ir/IR.scala:25-39: a balanced tree of SeqIRs
        n = len(defs)
        if n == 1: return defs[0]
        if n == 2: return VarDef(SeqIR(defs[0], defs[1]))
        k = n // 2
        return VarDef(SeqIR(SeqIR.of(defs[:k]), SeqIR.of(defs[k:])))


def _ref_key(e):
    """the Ref form of an Expr as a hashable key: Const by value, Param / VarRef by symbol (ir/IR.scala case-class equality)"""
    if isinstance(e, Const): return ("c", struct.pack("<d", e.value)) if e.value != 0.0 else ("c", 0.0)
    if isinstance(e, Param): return ("p", e.sym)
    return ("v", e.sym)                                                        # VarRef, or a VarDef seen through ref()


class Translator:
    """compute/Translator.scala:5-188"""

    def __init__(self):
        self.binary: Dict[tuple, int] = {}
        self.unary: Dict[tuple, int] = {}
        self.reals: Dict[Real, object] = {}

    @staticmethod
    def ref(expr):
        return VarRef(expr.sym) if isinstance(expr, VarDef) else expr

    def toExpr(self, r: Real):
        hit = self.reals.get(r)
        if hit is not None:
            return Translator.ref(hit)
        if isinstance(r, Parameter): expr = Param(r.sym)
        elif isinstance(r, Constant): expr = self.constToExpr(r)
        elif isinstance(r, Unary): expr = self.unaryExpr(self.toExpr(r.original), r.op)
        elif isinstance(r, Line): expr = self.makeLine(r.ax, r.b, _MULTIPLY_RING)
        elif isinstance(r, LogLine): expr = self.makeLine(r.ax, _ONE, _POW_RING)
        elif isinstance(r, Pow): expr = self.binaryExpr(self.toExpr(r.base), self.toExpr(r.exponent), POW)
        elif isinstance(r, Compare): expr = self.binaryExpr(self.toExpr(r.left), self.toExpr(r.right), COMPARE)
        elif isinstance(r, Lookup): expr = self.lookupExpr(r)
        else: raise TypeError(type(r))
        self.reals[r] = expr
        return expr

    @staticmethod
    def constToExpr(c: Constant):
        if isinstance(c, Scalar): return Const(c.value)
        ms = c.maybeScalar
        return Const(ms) if ms is not None else Param(c.sym)

    def _memoize(self, cache, exprKeys, opKey, make):
        refKeys = [tuple(_ref_key(e) for e in k) for k in exprKeys]
        hit = None
        for k in refKeys:
            hit = cache.get((k, opKey))
            if hit is not None:
                break
        if hit is not None:
            if any(isinstance(e, VarDef) for e in exprKeys[0]):
                raise RuntimeError("VarRef was used before its VarDef")
            return VarRef(hit)
        vd = VarDef(make())
        cache[(refKeys[0], opKey)] = vd.sym
        return vd

    def unaryExpr(self, original, op):
        return self._memoize(self.unary, [[original]], op, lambda: UnaryIR(original, op))

    def binaryExpr(self, left, right, op):
        key = [left, right]
        # NB (reference behaviour, compute/Translator.scala:44-51): the reversed key is also tried for the NON-commutative
        # operators -- `if (op.isCommutative) List(key) else List(key, key.reverse)` -- so pow(a, b) and pow(b, a) would share
        # a symbol if both occurred; restated as written
        keys = [key] if op in _COMMUTATIVE else [key, key[::-1]]
        return self._memoize(self.binary, keys, op, lambda: BinaryIR(left, right, op))

    def lookupExpr(self, lookup: Lookup):
        tableExprs = [self.toExpr(t) for t in lookup.table]
        defs = [e for e in tableExprs if isinstance(e, VarDef)]
        index = self.toExpr(lookup.index)
        refs = [Translator.ref(e) for e in tableExprs]
        return SeqIR.of(defs + [VarDef(LookupIR(index, refs, lookup.low))])

    def makeLine(self, ax: Coefficients, b: Constant, ring):
        terms = [(x, Translator.constToExpr(a)) for x, a in ax.toList()]
        allTerms = terms if b.isZero else [(b, Const(1.0))] + terms
        lazy = []
        for x, a in allTerms:
            if isinstance(a, Const) and a.value == 1.0:
                lazy.append(lambda x=x: self.toExpr(x))
            elif isinstance(a, Const) and a.value == 2.0:
                lazy.append(lambda x=x: self.binaryExpr(self.toExpr(x), self.toExpr(x), ring.plus))
            else:
                lazy.append(lambda x=x, a=a: self.binaryExpr(self.toExpr(x), a, ring.times))
        if ring.useTree:
            return self.combineTree(lazy, ring)
        acc = lazy[0]()
        for t in lazy[1:]:
            acc = self.binaryExpr(acc, t(), ring.plus)
        return acc

    def combineTree(self, terms, ring):
        while len(terms) > 1:
            nxt = []
            for i in range(0, len(terms), 2):
                if i + 1 < len(terms):
                    nxt.append(lambda l=terms[i], r=terms[i + 1]: self.binaryExpr(l(), r(), ring.plus))
                else:
                    nxt.append(terms[i])
            terms = nxt
        return terms[0]()


class _Ring:
    def __init__(self, times, plus, useTree): self.times, self.plus, self.useTree = times, plus, useTree
_MULTIPLY_RING = _Ring(MUL, ADD, False)
_POW_RING = _Ring(POW, MUL, True)


# ======================================================================================================= Target, inlining
def _leaves(real: Real):
    seen, params, cols = set(), [], []
    def loop(r: Real):
        if r in seen:
            return
        seen.add(r)
        if isinstance(r, Scalar): return
        if isinstance(r, Column): cols.append(r); return
        if isinstance(r, Parameter):
            params.append(r); loop(r.prior.density); return
        if isinstance(r, Unary): loop(r.original)
        elif isinstance(r, Line):
            for x, a in r.ax.toList():
                loop(x); loop(a)
            loop(r.b)
        elif isinstance(r, LogLine):
            for x, a in r.ax.toList():
                loop(x); loop(a)
        elif isinstance(r, Compare): loop(r.left); loop(r.right)
        elif isinstance(r, Pow): loop(r.base); loop(r.exponent)
        elif isinstance(r, Lookup):
            loop(r.index)
            for t in r.table: loop(t)
    loop(real)
    return params, cols


def findParameters(real: Real) -> List[Parameter]: return _leaves(real)[0]
def findColumns(real: Real) -> List[Column]: return _leaves(real)[1]


def inlinable(real: Real) -> bool:
    """TargetGroup.inlinable (compute/Target.scala:136-207): can the row sum be folded into the coefficients?"""
    seen: Dict[Real, tuple] = {}
    def merge(states): return (any(s[0] for s in states), any(s[1] for s in states), any(s[2] for s in states))
    def nonlinear(s): return (s[0], s[1], s[0] and s[1])
    def loop(r: Real):
        if r in seen:
            return seen[r]
        if isinstance(r, Scalar): res = (False, False, False)
        elif isinstance(r, Column): res = (False, True, False)
        elif isinstance(r, Parameter): res = (True, False, False)
        elif isinstance(r, Unary): res = nonlinear(loop(r.original))
        elif isinstance(r, Line):
            res = merge([loop(r.b)] + [s for x, a in r.ax.toList() for s in (loop(x), loop(a))])
        elif isinstance(r, LogLine):
            ts = [nonlinear(merge([loop(x), loop(a)])) for x, a in r.ax.toList()]
            st = merge(ts)
            if st[2] or not (st[0] and st[1]): res = st
            else: res = nonlinear(st) if any(t[0] and t[1] for t in ts) else st
        elif isinstance(r, Compare): res = nonlinear(merge([loop(r.left), loop(r.right)]))
        elif isinstance(r, Pow): res = nonlinear(merge([loop(r.base), loop(r.exponent)]))
        elif isinstance(r, Lookup):
            ist = loop(r.index)
            st = merge([merge([loop(t) for t in r.table]), ist])
            res = nonlinear(st) if ist[0] else st
        else: raise TypeError(type(r))
        seen[r] = res
        return res
    return not loop(real + real)[2]          # real + real triggers a distribute() if warranted


class PartialEvaluator:
    """compute/PartialEvaluator.scala:3-98: substitute row `rowIndex` of every Column and re-simplify"""

    def __init__(self, noChange: set, rowIndex: int):
        self.noChange, self.rowIndex, self.cache = noChange, rowIndex, {}

    def next(self): return PartialEvaluator(self.noChange, self.rowIndex + 1)

    def apply(self, real: Real):
        if real in self.noChange:
            return real, False
        hit = self.cache.get(real)
        if hit is not None:
            return hit, True
        v, changed = self.eval(real)
        if changed: self.cache[real] = v
        else: self.noChange.add(real)
        return v, changed

    def eval(self, real: Real):
        if isinstance(real, Scalar): return real, False
        if isinstance(real, Column): return Scalar(float(real.values[self.rowIndex])), True
        if isinstance(real, Line):
            terms = [(self.apply(x), self.apply(a)) for x, a in real.ax.toList()]
            b, bm = self.apply(real.b)
            if any(m1 or m2 for (_, m1), (_, m2) in terms) or bm:
                return Real.sum([x * a for (x, _), (a, _) in terms]) + b, True
            return real, False
        if isinstance(real, LogLine):
            terms = [(self.apply(x), self.apply(a)) for x, a in real.ax.toList()]
            if any(m1 or m2 for (_, m1), (_, m2) in terms):
                prod = None
                for (x, _), (a, _) in terms:
                    p = x.pow(a)
                    prod = p if prod is None else prod * p
                return prod, True
            return real, False
        if isinstance(real, Unary):
            r, m = self.apply(real.original)
            return (RealOps.unary(r, real.op), True) if m else (real, False)
        if isinstance(real, Compare):
            l, lm = self.apply(real.left); r, rm = self.apply(real.right)
            return (RealOps.compare(l, r), True) if (lm or rm) else (real, False)
        if isinstance(real, Pow):
            b, bm = self.apply(real.base); e, em = self.apply(real.exponent)
            return (b.pow(e), True) if (bm or em) else (real, False)
        if isinstance(real, Lookup):
            i, im = self.apply(real.index)
            nt = [self.apply(t) for t in real.table]
            if im or any(m for _, m in nt):
                return Lookup.apply(i, [t for t, _ in nt], real.low), True
            return real, False
        if isinstance(real, Parameter): return real, False
        raise TypeError(type(real))

    @staticmethod
    def inline(real: Real, nRows: int) -> Real:
        acc, pe = Real.zero, PartialEvaluator(set(), 0)
        for _ in range(nRows):
            acc = acc + pe.apply(real)[0]
            pe = pe.next()
        return acc


class Target:
    """compute/Target.scala:5-39"""

    def __init__(self, name: str, real: Real, parameters: List[Parameter], inline: bool = True):
        columns = findColumns(real)
        nRows = columns[0].values.size if columns else 0
        if inline and nRows > 0 and inlinable(real):
            real, columns = PartialEvaluator.inline(real, nRows), []
        self.name, self.real, self.columns = name, real, columns
        self.gradient = Gradient.derive(parameters, real) if parameters else []
        seen, gcols = set(id(c) for c in columns), []
        for g in self.gradient:
            for c in findColumns(g):
                if id(c) not in seen:
                    seen.add(id(c)); gcols.append(c)
        self.gradientColumns = gcols


class TargetGroup:
    """compute/Target.scala:41-84.  inline = False keeps every likelihood un-inlined (streamed over its rows): what bench.py
    measures on cfg 2; the reference itself would inline that model (3 covariates: 15 distributed terms < 20)."""

    def __init__(self, reals: List[Real], track: Sequence[Real] = (), inline: bool = True):
        params, seen = [], set()
        for r in list(reals) + list(track):
            for p in findParameters(r):
                if id(p) not in seen:
                    seen.add(id(p)); params.append(p)
        self.parameters = sorted(params, key=lambda p: p.sym)
        priors, pseen = [], set()
        for p in self.parameters:                                   # parameters.map(_.prior).toSet (insertion order)
            if id(p.prior) not in pseen:
                pseen.add(id(p.prior)); priors.append(p.prior)
        dens, dseen = [], set()
        for pr in priors:                                           # priors.map(_.density): a Set[Real] (equal densities collapse)
            if pr.density not in dseen:
                dseen.add(pr.density); dens.append(pr.density)
        self.targets = [Target("prior", Real.sum(dens), self.parameters, inline)]
        self.targets += [Target("t_%d" % i, r, self.parameters, inline) for i, r in enumerate(reals)]
        self.columns = [c for t in self.targets for c in (t.columns + t.gradientColumns)]
        self.data = [[c.values for c in (t.columns + t.gradientColumns)] for t in self.targets]
        self.outputs = [(name, r) for t in self.targets
                        for name, r in [(t.name, t.real)] + [("%s_grad_%d" % (t.name, i), g) for i, g in enumerate(t.gradient)]]


# ======================================================================================================= RIR writer
def to_rir(group: TargetGroup, kind: int = 0):
    """HipCompiler.compileTargets (integration/scala/HipCompiler.scala) in Python: the inputs / outputs Compiler.compile
    hands to ir.CompiledFunction (compute/Compiler.scala:22-30), translated by ONE Translator in output order and written
    as RIR.  Returns (rir bytes, columns flattened target-major, rows per target, n_params)."""
    nParams = len(group.parameters)
    inputIndex = {p.sym: i for i, p in enumerate(group.parameters)}
    for j, c in enumerate(group.columns):
        inputIndex[c.sym] = nParams + j
    tr = Translator()
    exprs = [tr.toExpr(r) for _, r in group.outputs]
    ids: Dict[int, int] = {}
    nodes = bytearray()
    count = [0]

    def emit(words, const=None) -> int:
        nonlocal nodes
        nodes += struct.pack("<%dI" % len(words), *[w & 0xFFFFFFFF for w in words])
        if const is not None:
            nodes += struct.pack("<d", const)
        count[0] += 1
        return count[0] - 1

    def ref(e) -> int:
        if isinstance(e, Const):
            if math.isnan(e.value):
                raise ArithmeticError("NaN constant")
            return emit([0], e.value)
        if isinstance(e, Param): return emit([1, inputIndex[e.sym]])
        if isinstance(e, VarRef): return ids[e.sym]
        rhs = e.rhs
        if isinstance(rhs, BinaryIR):
            a = ref(rhs.left); b = ref(rhs.right); i = emit([_RIR_BINARY[rhs.op], a, b])
        elif isinstance(rhs, UnaryIR):
            a = ref(rhs.original); i = emit([_RIR_UNARY[rhs.op], a])
        elif isinstance(rhs, LookupIR):
            a = ref(rhs.index); ts = [ref(t) for t in rhs.table]
            i = emit([18, a, rhs.low, len(ts)] + ts)
        elif isinstance(rhs, SeqIR):
            a = ref(rhs.first); b = ref(rhs.second); i = emit([19, a, b])
        else:
            raise TypeError(type(rhs))
        ids[e.sym] = i
        return i

    outIds = [ref(e) for e in exprs]
    perTarget = 1 + nParams
    nCols = [len(cs) for cs in group.data]
    out = bytearray(struct.pack("<6I", 0x31524952, 1, nParams, len(nCols), count[0], kind))
    for t, nc in enumerate(nCols):
        out += struct.pack("<2I", nc, 0)
        out += struct.pack("<%dI" % perTarget, *outIds[t * perTarget:(t + 1) * perTarget])
    out += nodes
    columns = [c for cs in group.data for c in cs]
    rows = [int(cs[0].size) if cs else 0 for cs in group.data]
    return bytes(out), columns, rows, nParams


def requirements_rir(parameters: Sequence[Parameter], reqs: Sequence[Real]) -> bytes:
    """HipCompiler.compileRequirements: kind 1, one data-free target per requirement, gradient slots = constant 0"""
    class _G: pass
    g = _G()
    g.parameters, g.columns = list(parameters), []
    g.data = [[] for _ in reqs]
    g.outputs = [(str(i), r) for i, q in enumerate(reqs) for r in [Real.of(q)] + [Real.zero] * len(parameters)]
    return to_rir(g, kind=1)[0]


# ======================================================================================================= Evaluator
def evaluate(r: Real, params: Dict[Parameter, object] = None, memo=None):
    """Evaluator (compute/Evaluator.scala:5-48): numeric value of a Real; columns evaluate to arrays (row-wise)"""
    memo = {} if memo is None else memo
    hit = memo.get(r)
    if hit is not None:
        return hit
    with np.errstate(all="ignore"):
        if isinstance(r, Scalar): v = np.float64(r.value)
        elif isinstance(r, Column): v = r.values
        elif isinstance(r, Parameter): v = np.asarray(params[r], dtype=np.float64)
        elif isinstance(r, Line):
            v = evaluate(r.b, params, memo)
            # Evaluator: l.ax.foldLeft(b) { acc + toDouble(x) * a }
            for x, a in r.ax.toList():
                v = v + evaluate(x, params, memo) * evaluate(a, params, memo)
        elif isinstance(r, LogLine):
            v = np.float64(1.0)
            for x, a in r.ax.toList():
                v = v * _java_pow(evaluate(x, params, memo), evaluate(a, params, memo))
        elif isinstance(r, Unary):
            fn = {EXP: np.exp, LOG: np.log, ABS: np.abs, NOOP: lambda t: t, SIN: np.sin, COS: np.cos, TAN: np.tan, ASIN: np.arcsin,
                  ACOS: np.arccos, ATAN: np.arctan}[r.op]
            v = fn(evaluate(r.original, params, memo))
        elif isinstance(r, Compare):
            a, b = evaluate(r.left, params, memo), evaluate(r.right, params, memo)
            v = np.where(a > b, 1.0, np.where(a == b, 0.0, -1.0))
        elif isinstance(r, Pow):
            v = _java_pow(evaluate(r.base, params, memo), evaluate(r.exponent, params, memo))
        elif isinstance(r, Lookup):
            idx = evaluate(r.index, params, memo)
            tab = [evaluate(t, params, memo) for t in r.table]
            k0 = np.trunc(np.nan_to_num(idx)).astype(np.int64) - r.low
            if np.any((k0 < 0) | (k0 >= len(tab))):
                raise IndexError("Lookup index out of range")
            v = tab[0]
            for j in range(1, len(tab)):
                v = np.where(k0 == j, tab[j], v)
        else:
            raise TypeError(type(r))
    memo[r] = v
    return v
