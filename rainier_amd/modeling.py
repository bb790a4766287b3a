"""A small restatement of Rainier's modelling surface, enough to write the reference's own test models
the way the reference writes them:

    mu = Normal(0, 10).latent ; sigma = Uniform(0, 1).latent
    model = Model.observe([1.0, 2.0, 3.0], Normal(mu, sigma))          # optimizer/OptimizerTest.scala:8-13
    SBC([Uniform(0, 1)], lambda x: Normal(x, 1))                         # core/SBCModel.scala:46-47

It mirrors, name for name, the pieces of rainier-core that sit in front of the hot path:
  Continuous / StandardContinuous / LocationScaleFamily, Normal, Cauchy, Laplace, Gamma, Exponential, Beta, LogNormal,
  Uniform (core/Continuous.scala); Scale / Translate / Exp injections (core/Injection.scala); the Supports
  (core/Support.scala); Bernoulli, Geometric, NegativeBinomial, Poisson, Binomial (core/Discrete.scala, Multinomial.scala);
  Combinatorics (Nemes' log-Gamma); Model.observe (core/Model.scala:52-75); SBC.synthesize / fit (core/SBC.scala:61-69).
`Real` is rainier-compute's own algebra as restated in compute.py (Line / LogLine normal forms, Gradient on the Real DAG,
TargetGroup.inlinable + PartialEvaluator.inline, Translator): `Model.compile()` produces the RIR that
Compiler.compileTargets would hand to the reference's back end for the same model -- including the 8-way split of
Model.observe (core/Model.scala:71-132) and the compile-time folding of inlinable likelihoods into O(1) targets.

This is a TEST / AUTHORING aid (the product boundary is the C ABI; a JVM deployment keeps Rainier's own front-end).
Generators take any object with `next_double()` / `next_gaussian()` (java.util.Random semantics): the tests pass the
oracle's bit-exact stream; `JavaRandom` below is a pure-Python stand-in (its gaussians use libm's log).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence

import numpy as np

from . import compute as _c
from . import models as _models
from .compute import Bounds, Real, evaluate  # noqa: F401  (Real is rainier-compute's algebra, restated in compute.py)


# ---------------------------------------------------------------------------------------------------- RNG
class JavaRandom:
    """java.util.Random (the stream behind ScalaRNG, sampler/RNG.scala:20-26); gaussians via libm log/sqrt."""

    def __init__(self, seed: int):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1); self.nn = None
    def _next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        return self.seed >> (48 - bits)
    def next_double(self): return ((self._next(26) << 27) + self._next(27)) * 2.0 ** -53
    def next_gaussian(self):
        if self.nn is not None:
            g, self.nn = self.nn, None
            return g
        while True:
            v1 = 2 * self.next_double() - 1; v2 = 2 * self.next_double() - 1; s = v1 * v1 + v2 * v2
            if 0 < s < 1:
                break
        m = math.sqrt(-2 * math.log(s) / s)
        self.nn = v2 * m
        return v1 * m


def _d(x) -> float:
    return float(evaluate(Real.of(x)))


# ---------------------------------------------------------------------------------------------------- Combinatorics
class Combinatorics:
    """core/Combinatorics.scala:10-37 (log of the function named)."""

    @staticmethod
    def gamma(z):
        z = Real.of(z)
        if z == Real.zero: return Real.infinity
        if z == Real.one or z == Real.two: return Real.zero
        v = z + 1
        w = v + (Real.of(1.0) / ((12 * v) - (Real.of(1.0) / (10 * v))))
        return (Real.of(math.pi * 2).log() / 2) - (v.log() / 2) + (v * (w.log() - 1)) - z.log()
    @staticmethod
    def factorial(k): return Combinatorics.gamma(Real.of(k) + 1)
    @staticmethod
    def beta(a, b): return Combinatorics.gamma(a) + Combinatorics.gamma(b) - Combinatorics.gamma(Real.of(a) + b)


# ---------------------------------------------------------------------------------------------------- supports, injections
class UnboundedSupport:
    def transform(self, v): return v
    def logJacobian(self, v): return Real.of(0.0)

class BoundedSupport:
    def __init__(self, lo, hi): self.lo, self.hi = Real.of(lo), Real.of(hi)
    def transform(self, v): return v.logistic * (self.hi - self.lo) + self.lo
    def logJacobian(self, v): return v.logistic.log() + (1 - v.logistic).log() + (self.hi - self.lo).log()

class BoundedBelowSupport:
    def __init__(self, lo=0.0): self.lo = Real.of(lo)
    def transform(self, v): return v.exp() + self.lo
    def logJacobian(self, v): return v


class Continuous:
    """core/Continuous.scala:10-22."""
    support = None
    def logDensity(self, x) -> Real: raise NotImplementedError
    def generate(self, rng) -> float: raise NotImplementedError
    @property
    def latent(self) -> Real: raise NotImplementedError
    def latentVec(self, k: int): return [self.latent for _ in range(k)]     # Vec.from(List.fill(k)(latent))
    def scale(self, a): return _Scaled(self, Real.of(a))
    def translate(self, b): return _Translated(self, Real.of(b))
    def exp(self): return _Exped(self)


class StandardContinuous(Continuous):
    @property
    def latent(self) -> Real:          # core/Continuous.scala:27-35
        x = Real.parameter(lambda x: self.support.logJacobian(x) + self.logDensity(self.support.transform(x)))
        return self.support.transform(x)


class _Scaled(Continuous):             # Scale(a).transform(dist), core/Injection.scala:48-66
    def __init__(self, d, a): self.d, self.a = d, a
    def logDensity(self, y): return self.d.logDensity(Real.of(y) / self.a) + self.a.log() * -1
    def generate(self, rng): return self.d.generate(rng) * _d(self.a)
    @property
    def latent(self): return self.d.latent * self.a

class _Translated(Continuous):         # Translate(b), core/Injection.scala:68-86
    def __init__(self, d, b): self.d, self.b = d, b
    def logDensity(self, y): return self.d.logDensity(Real.of(y) - self.b) + Real.of(0.0)
    def generate(self, rng): return self.d.generate(rng) + _d(self.b)
    @property
    def latent(self): return self.d.latent + self.b

class _Exped(Continuous):              # Exp, core/Injection.scala:88-110
    def __init__(self, d): self.d = d
    def logDensity(self, y):
        y = Real.of(y)
        return Real.gt(y, Real.zero, self.d.logDensity(y.log()) + y.log() * -1, Real.negInfinity)
    def generate(self, rng): return math.exp(self.d.generate(rng))
    @property
    def latent(self): return self.d.latent.exp()


class _LocationScale:
    class _Std(StandardContinuous):
        support = UnboundedSupport()
        def __init__(self, fam): self.fam = fam
        def logDensity(self, x): return self.fam._logDensity(Real.of(x))
        def generate(self, rng): return self.fam._generate(rng)
    def __init__(self): self.standard = _LocationScale._Std(self)
    def __call__(self, location, scale): return self.standard.scale(scale).translate(location)

class _Normal(_LocationScale):         # core/Continuous.scala:63-67
    def _logDensity(self, x): return ((x * x) / -2.0) - 0.5 * Real.of(2 * math.pi).log()
    def _generate(self, rng): return rng.next_gaussian()
class _Cauchy(_LocationScale):         # :72-77
    def _logDensity(self, x): return (((x * x) + 1) * math.pi).log() * -1
    def _generate(self, rng): return rng.next_gaussian() / rng.next_gaussian()
class _Laplace(_LocationScale):        # :82-90
    def _logDensity(self, x): return Real.of(0.5).log() - x.abs()
    def _generate(self, rng):
        u = rng.next_double() - 0.5
        return ((u > 0) - (u < 0)) * -1 * math.log(1 - (2 * abs(u)))
Normal, Cauchy, Laplace = _Normal(), _Cauchy(), _Laplace()


class _GammaStandard(StandardContinuous):   # core/Continuous.scala:103-147
    support = BoundedBelowSupport(0.0)
    def __init__(self, shape): self.shape = Real.of(shape)
    def logDensity(self, x):
        x = Real.of(x)
        return Bounds.positive(x, lambda: (self.shape - 1) * x.log() - Combinatorics.gamma(self.shape) - x)
    def generate(self, rng):
        a = _d(self.shape)
        if a < 1:
            u = rng.next_double()
            return self._mt(a + 1, rng) * math.pow(u, 1.0 / a)
        return self._mt(a, rng)
    @staticmethod
    def _mt(a, rng):                       # Marsaglia-Tsang
        d = a - 1.0 / 3.0; c = (1.0 / 3.0) / math.sqrt(d)
        while True:
            x = rng.next_gaussian(); v = 1.0 + c * x
            while v <= 0:
                x = rng.next_gaussian(); v = 1.0 + c * x
            v3 = v * v * v; u = rng.next_double()
            if (u < 1 - 0.0331 * x * x * x * x) or (math.log(u) < 0.5 * x * x + d * (1 - v3 + math.log(v3))):
                return d * v3

class Gamma:
    @staticmethod
    def standard(shape): return _GammaStandard(shape)
    def __new__(cls, shape, scale): return _GammaStandard(shape).scale(scale)

def Exponential(rate): return _GammaStandard(1.0).scale(Real.of(1.0) / Real.of(rate))     # :152-158
def LogNormal(location, scale): return Normal(location, scale).exp()                      # :194-197

class Beta(StandardContinuous):            # :163-189
    support = BoundedSupport(0.0, 1.0)
    def __init__(self, a, b): self.a, self.b = Real.of(a), Real.of(b)
    def logDensity(self, u):
        u = Real.of(u)
        return Bounds.zeroToOne(u, lambda: (self.a - 1) * u.log() + (self.b - 1) * (1 - u).log() - Combinatorics.beta(self.a, self.b))
    def generate(self, rng):
        z1 = _GammaStandard(self.a).generate(rng); z2 = _GammaStandard(self.b).generate(rng)
        return z1 / (z1 + z2)

class Mixture(Continuous):                 # core/Continuous.scala:218-248 (logDensity; components: list of (dist, weight))
    def __init__(self, components): self.components = [(d, Real.of(w)) for d, w in components]
    def logDensity(self, x):
        x = Real.of(x)
        return Real.logSumExp([d.logDensity(x) + w.log() for d, w in self.components])


class _UniformStandard(StandardContinuous):  # :202-213
    support = BoundedSupport(0.0, 1.0)
    def logDensity(self, x): return Beta(1, 1).logDensity(x)
    def generate(self, rng): return rng.next_double()
def Uniform(lo, hi): return _UniformStandard().scale(Real.of(hi) - Real.of(lo)).translate(lo)


# ---------------------------------------------------------------------------------------------------- discrete
class Discrete:
    def logDensity(self, v) -> Real: raise NotImplementedError
    def generate(self, rng) -> float: raise NotImplementedError

class Bernoulli(Discrete):                 # core/Discrete.scala:38-52
    def __init__(self, p): self.p = Real.of(p)
    def logDensity(self, v): return Real.eq(v, 0.0, (1 - self.p).log(), self.p.log())
    def generate(self, rng): return 1.0 if rng.next_double() <= _d(self.p) else 0.0

class Geometric(Discrete):                 # :59-73
    def __init__(self, p): self.p = Real.of(p)
    def logDensity(self, v): return self.p.log() + Real.of(v) * (1 - self.p).log()
    def generate(self, rng):
        u = rng.next_double(); q = _d(self.p)
        return float(math.floor(math.log(u) / math.log(1 - q)))

class NegativeBinomial(Discrete):          # :81-120
    def __init__(self, p, n): self.p, self.n = Real.of(p), Real.of(n)
    def logDensity(self, v):
        v = Real.of(v); n, p = self.n, self.p
        return (Combinatorics.factorial(n + v - 1) - Combinatorics.factorial(v) - Combinatorics.factorial(n - 1)
                + n * (1 - p).log() + v * p.log())
    def generate(self, rng):
        p, n = _d(self.p), _d(self.n)
        if p < -100 / n + 1 and p > 100 / n - .25:
            return float(max(int(Normal(n * p / (1 - p), math.sqrt(n * p) / (1 - p)).generate(rng)), 0))
        g = Geometric(1 - self.p)
        return float(sum(g.generate(rng) for _ in range(int(n))))

class Poisson(Discrete):                   # :127-189
    def __init__(self, lam): self.lam = Real.of(lam)
    def logDensity(self, v): return self.lam.log() * v - self.lam - Combinatorics.factorial(v)
    def generate(self, rng):
        lam = _d(self.lam)
        if lam < 30.0:
            l = math.exp(-lam)
            if l >= 1.0:
                return 0.0
            k, p = 0, 1.0
            while p > l:
                k += 1; p *= rng.next_double()
            return float(k - 1)
        c = 0.767 - 3.36 / lam; beta = math.pi / math.sqrt(3.0 * lam); alpha = beta * lam
        kk = math.log(c) - lam - math.log(beta)
        while True:
            u = rng.next_double()
            x = (alpha - math.log((1.0 - u) / u)) / beta
            n = int(math.floor(x + 0.5))
            if n >= 0:
                v = rng.next_double(); y = alpha - beta * x
                lhs = y + math.log(v / math.pow(1.0 + math.exp(y), 2))
                xx = float(n + 1)
                rhs = kk + n * math.log(lam) - (((xx - 0.5) * math.log(xx)) - xx + (0.5 * math.log(2 * math.pi)))
                if lhs <= rhs:
                    return float(n)

class Binomial(Discrete):                  # :196-232 + Multinomial.scala:17-29
    def __init__(self, p, k): self.p, self.k = Real.of(p), Real.of(k)
    def logDensity(self, v):
        v = Real.of(v); terms = []
        for i, p in ((v, self.p), (self.k - v, 1 - self.p)):
            terms.append(Real.eq(i, 0.0, 0.0, i * p.log()) - Combinatorics.factorial(i))
        return Combinatorics.factorial(self.k) + Real.sum(terms)
    def generate(self, rng):
        p, k = _d(self.p), _d(self.k)
        if k >= 100 and k * p <= 10:
            return float(min(Poisson(p * k).generate(rng), int(k)))
        if k >= 100 and k * p >= 9 and k * (1.0 - p) >= 9:
            return float(min(max(int(Normal(k * p, math.sqrt(k * p * (1 - p))).generate(rng)), 0), int(k)))
        return float(sum(1 for _ in range(int(k)) if p >= rng.next_double()))   # categorical cdf(true) = p >= u


# ---------------------------------------------------------------------------------------------------- Model, SBC
NumSplits = 8   # core/Model.scala:98


def _split(ts):
    """Model.split (core/Model.scala:117-132): an initial chunk + NumSplits equal chunks"""
    ts = list(ts)
    splitSize = int((len(ts) - 1) / NumSplits)   # Scala's Int division truncates toward zero: an empty list gives 0, not -1
    initSize = len(ts) - splitSize * NumSplits
    if splitSize == 0:
        return ts[:initSize], []
    return ts[:initSize], [ts[initSize + i * splitSize: initSize + (i + 1) * splitSize] for i in range(NumSplits)]


def _column_density(dist, values) -> Real:
    """Distribution.logDensity(seq) = Vec.from(seq).map(logDensity).columnize (core/Continuous.scala:14): the density of one
    Column holding the observations"""
    if len(values) == 1:      # the index column [0.0] has a maybeScalar, so Lookup.apply picks the (scalar) entry (compute/Real.scala:310-318)
        return dist.logDensity(Real.of(float(values[0])))
    return dist.logDensity(Real.doubles(values))


class Model:
    """core/Model.scala:7-75: a list of likelihood Reals over shared parameters (+ tracked Reals)."""

    def __init__(self, likelihoods, track=()):
        self.likelihoods = list(likelihoods)
        self.track = list(track)
        self._group = {}

    @staticmethod
    def observe(values: Sequence[float], dist, split: bool = True) -> "Model":
        """Model.observe(ys, lh) (core/Model.scala:74-82): the observations are cut into an initial chunk and 8 equal chunks;
        the model has the likelihoods [lh(init), sum of lh(chunk_i)] -- the second one reads 8 columns per row.
        split = False: one likelihood over one column (the form the engine streams for the BASELINE configurations)."""
        if not split:
            return Model([_column_density(dist, values)])
        init, splits = _split(values)
        initReal = _column_density(dist, init)
        if not splits:
            return Model([initReal])
        return Model([initReal, Real.sum([_column_density(dist, sp) for sp in splits])])

    @staticmethod
    def observe_vec(values: Sequence[float], covariates: Sequence[Sequence[float]], fn, split: bool = True) -> "Model":
        """Model.observe(ys, lhs: Vec[D]) (core/Model.scala:84-96) for lhs = Vec.from(xs).map { case (u, v, ...) => dist }:
        `columnize` turns every covariate into a Column (the index Lookup over constant tables folds, compute/Real.scala:310-325),
        so each chunk is  fn(column_u, column_v, ...).logDensity(column_y)  -- the README regression's shape.
        covariates: one sequence per covariate (column-major)."""
        covs = [np.asarray(c, dtype=np.float64) for c in covariates]
        ys = np.asarray(values, dtype=np.float64)
        def chunk(idx):
            return fn(*[Real.doubles(c[idx]) for c in covs]).logDensity(Real.doubles(ys[idx]))
        n = len(ys)
        if not split:
            return Model([chunk(slice(0, n))])
        init, splits = _split(range(n))
        initReal = chunk(slice(0, len(init)))
        if not splits:
            return Model([initReal])
        return Model([initReal, Real.sum([chunk(slice(sp[0], sp[-1] + 1)) for sp in splits])])

    @staticmethod
    def track(reals) -> "Model":
        """Model.track(track) = new Model(List(Real.zero), track) (core/Model.scala:67)"""
        return Model([Real.zero], list(reals))

    def merge(self, other: "Model") -> "Model":
        return Model(self.likelihoods + other.likelihoods, self.track + other.track)

    def targetGroup(self, inline: bool = True) -> "_c.TargetGroup":
        if inline not in self._group:
            self._group[inline] = _c.TargetGroup(self.likelihoods, self.track, inline)
        return self._group[inline]

    def parameters(self) -> List[Real]:
        return self.targetGroup().parameters

    def compile(self, name: str = "model", inline: bool = True) -> "_models.ModelSpec":
        """Model.dataFn = Compiler.default.compileTargets(targetGroup) (core/Model.scala:32-34), with the HIP serialiser in
        place of the ASM back end: target 0 = "prior", then one target per likelihood, outputs = value :: gradient.
        inline = False skips PartialEvaluator.inline so that every likelihood is streamed over its rows."""
        rir, columns, rows, n = _c.to_rir(self.targetGroup(inline))
        return _models.ModelSpec(name, rir, columns, rows, n, {"kind": "modeling", "inline": inline})

    def predict(self, real, draws: np.ndarray) -> np.ndarray:
        """Trace.predict(real) for draws [..., nVars] (host evaluation of the tracked Real)."""
        params = self.parameters()
        d = np.asarray(draws, dtype=np.float64)
        return np.asarray(evaluate(Real.of(real), {p: d[..., i] for i, p in enumerate(params)}))


class SBC:
    """core/SBC.scala:14-69: priors + a function from the prior draws to the likelihood (the summary is the first prior)."""

    def __init__(self, priors: Sequence[Continuous], fn: Callable[..., object]):
        self.priors, self.fn = list(priors), fn

    def synthesize(self, samples: int, rng):
        """(values, true value): prior draws first, then `samples` draws of the likelihood, all from the one stream."""
        truth = [p.generate(rng) for p in self.priors]
        dist = self.fn(*[Real.of(t) for t in truth])
        return [dist.generate(rng) for _ in range(samples)], truth[0]

    def fit(self, values, split: bool = True):
        latents = [p.latent for p in self.priors]
        return Model.observe(values, self.fn(*latents), split), latents[0]
