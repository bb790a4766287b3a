"""End-to-end known answer from the reference's own regression suite: SBCTest / SBCUniformNormal.

rainier-test/.../core/SBCTest.scala:7-21 requires `model.sample(...).predict(x)` to equal the stored goldset
(SBCModel.scala:46-60) to RELATIVE 1e-10.  The goldset was produced by the real JVM stack, so reproducing it pins,
in one go: the java.util.Random stream and its consumption order (synthesize: 1 uniform + 1000 gaussians, then
LeapFrog.initialize, per-iteration momenta + accept uniforms), LeapFrog, HMCSampler(1), DualAvgTuner /
findReasonableStepSize and the Driver loop (10000 warm-up + 30 sampling iterations).

The model is hand-derived from the reference front-end (no JVM here):
  prior   Uniform(0,1).latent : parameter v, x = logistic(v) = 1/(1+exp(-v))            (core/Support.scala:62-68)
          density log(x) + log(1-x) (+ log(1-0) = 0) + Beta(1,1).logDensity = 0           (core/Continuous.scala:160-200)
  data    y_i = g_i * 1.0 + x0, x0 = first nextDouble, g_i the next 1000 nextGaussian    (core/SBC.scala:61-69, Injection.scala:31-36)
  lik     sum_i -(y_i - x)^2/2 - 0.5 log 2 pi                                             (core/Continuous.scala:63-67)
The reference folds the 1000 rows into sufficient statistics before compiling (compute/PartialEvaluator.scala); we
stream them instead, so the floating-point evaluation order differs -- which is exactly why the reference's own
tolerance is 1e-10 rather than bit equality.
"""
import json
import os

import numpy as np
import pytest

from rainier_amd import models
from rainier_amd.frontend import Graph
from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "sbc_uniform_normal.json")))


def sbc_uniform_normal_spec():
    rng = O.JavaRandom(GOLD["seed"])
    x0 = rng.next_double()                                   # Uniform(0,1).generator
    ys = np.array([rng.next_gaussian() * 1.0 + x0 for _ in range(GOLD["synthetic_samples"])])
    g = Graph(1, [0, 1])
    v = g.param(0)
    x = 1.0 / ((v * -1.0).exp() + 1.0)
    prior = x.log() + (1.0 - x).log()
    y = g.col(1, 0)
    d = y - x
    row = (d * d) / -2.0 - models.HALF_LOG_2PI
    spec = models.ModelSpec("sbc_uniform_normal", g.compile([prior, row]), [ys], [0, len(ys)], 1)
    return spec, rng.r, x0


def predict(draws):
    return 1.0 / (1.0 + np.exp(-draws[:, 0]))


def test_oracle_reproduces_reference_sbc_goldset(oracle):
    spec, rstate, x0 = sbc_uniform_normal_spec()
    assert 0.0 < x0 < 1.0
    gold = np.array(GOLD["goldset"])
    cfg = O.make_config(sampler=O.HMC, n_steps=1, iterations=len(gold), warmup=GOLD["warmup"], step_tuner=O.STEP_DUALAVG,
                        delta=0.8, mass_tuner=O.MASS_IDENTITY, math_mode=O.JM_LIBM)
    d = O.OracleDensity(spec, O.JM_LIBM)
    # Driver.sample continues the SAME ScalaRNG stream that synthesize consumed from
    got = predict(sample_chain_with_rng_state(oracle, d, cfg, rstate))
    rel = np.abs((got - gold) / gold)
    assert rel.max() < 1e-10, rel       # SBCTest.scala:9  val Epsilon = 1e-10
    # the deterministic-math mode the GPU engine implements agrees with it as well
    cfg.math_mode = O.JM_DET
    spec2, rstate2, _ = sbc_uniform_normal_spec()
    got2 = predict(sample_chain_with_rng_state(oracle, O.OracleDensity(spec2, O.JM_DET), cfg, rstate2))
    assert np.abs((got2 - gold) / gold).max() < 1e-10


def sample_chain_with_rng_state(oracle, density, cfg, rstate):
    """orc_sample_chain seeds its own java.util.Random; to continue an existing stream we invert the seed scramble:
    Random(s).seed = (s ^ 0x5DEECE66D) & mask  =>  s = state ^ 0x5DEECE66D."""
    assert not rstate.have_next        # 1000 gaussians = 500 pairs: no cached value pending
    seed = rstate.seed ^ 0x5DEECE66D
    draws, mass, st, rc = O.sample_chain(density.fn_ptr, density.handle, 1, cfg, seed)
    assert rc == 0
    return draws


# ---- more of the reference's goldsets (SBCModel.scala:46-267 via tests/golden/sbc_goldsets.json) -----------------
ALL = json.load(open(os.path.join(HERE, "golden", "sbc_goldsets.json")))


def _uniform01_prior(g):
    v = g.param(0)
    x = 1.0 / ((v * -1.0).exp() + 1.0)          # BoundedSupport(0,1).transform = logistic (core/Support.scala:62-68)
    return x, x.log() + (1.0 - x).log()         # logJacobian; Beta(1,1).logDensity folds to 0


def bernoulli_spec():
    """SBCBernoulli: Bernoulli(x), x ~ Uniform(0,1).  Data: u <= x0 ? 1 : 0 (core/Discrete.scala:43-48);
    logDensity = Real.eq(v, 0, log(1-p), log p) (Discrete.scala:50-51) -> Lookup(Compare) on the engine."""
    rng = O.JavaRandom(ALL["seed"]); x0 = rng.next_double()
    ys = np.array([1.0 if rng.next_double() <= x0 else 0.0 for _ in range(1000)])
    g = Graph(1, [0, 1]); x, prior = _uniform01_prior(g); y = g.col(1, 0)
    row = g.eq(y, 0.0, (1.0 - x).log(), x.log())
    return models.ModelSpec("sbc_bernoulli", g.compile([prior, row]), [ys], [0, 1000], 1), rng.r, lambda d: 1 / (1 + np.exp(-d[:, 0]))


def geometric_spec():
    """SBCGeometric: data floor(log(u)/log(1-x0)) (Discrete.scala:64-69); logDensity = log p + v log(1-p) (:71-72)."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = rng.next_double()
    ys = np.array([float(math.floor(math.log(rng.next_double()) / math.log(1 - x0))) for _ in range(1000)])
    g = Graph(1, [0, 1]); x, prior = _uniform01_prior(g); y = g.col(1, 0)
    row = x.log() + y * (1.0 - x).log()
    return models.ModelSpec("sbc_geometric", g.compile([prior, row]), [ys], [0, 1000], 1), rng.r, lambda d: 1 / (1 + np.exp(-d[:, 0]))


def laplace_spec():
    """SBCLaplace: Laplace(x, x), x ~ LogNormal(0,1) = exp(z), z standard normal (Continuous.scala:63-67,194-197;
    Injection.scala:82-103).  Data: x0 = exp(g1); y = signum(u)(-1)log(1-2|u|) * x0 + x0, u = uniform - 0.5
    (Continuous.scala:83-90).  logDensity(y) = log 0.5 - |(y-x)/x| - log x.  The first gaussian leaves its twin cached, so
    the sampler's first standardNormal returns it: the stream is continued by STATE, not by seed."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = math.exp(rng.next_gaussian() * 1.0 + 0.0)
    ys = []
    for _ in range(1000):
        u = rng.next_double() - 0.5
        sgn = (u > 0) - (u < 0)
        ys.append((sgn * -1 * math.log(1 - (2 * abs(u)))) * x0 + x0)
    g = Graph(1, [0, 1]); z = g.param(0); x = z.exp(); y = g.col(1, 0)
    prior = models.std_normal_logpdf(z)
    row = (math.log(0.5) - ((y - x) / x).abs()) - x.log()
    return models.ModelSpec("sbc_laplace", g.compile([prior, row]), [np.array(ys)], [0, 1000], 1), rng.r, lambda d: np.exp(d[:, 0])


@pytest.mark.parametrize("name,builder", [("SBCBernoulli", bernoulli_spec), ("SBCGeometric", geometric_spec),
                                          ("SBCLaplace", laplace_spec)])
def test_oracle_reproduces_more_reference_goldsets(oracle, name, builder):
    spec, rstate, predict_fn = builder()
    gold = np.array(ALL["models"][name]["goldset"])
    for mode in (O.JM_LIBM, O.JM_DET):
        cfg = O.make_config(sampler=O.HMC, n_steps=1, iterations=len(gold), warmup=ALL["warmup"], step_tuner=O.STEP_DUALAVG,
                            delta=0.8, mass_tuner=O.MASS_IDENTITY, math_mode=mode)
        d = O.OracleDensity(spec, mode)
        draws, _, st, rc = O.sample_chain_state(d.fn_ptr, d.handle, 1, cfg, rstate)
        assert rc == 0
        rel = np.abs((predict_fn(draws) - gold) / gold)
        assert rel.max() < 1e-10, (name, mode, rel.max())


def lognormal_spec():
    """SBCLogNormal: LogNormal(x, x) with x ~ LogNormal(0,1) = exp(z).  LogNormal(l, s) = Normal(l, s).exp
    (Continuous.scala:194-197): logDensity(y) = Normal(l,s).logDensity(log y) - log y (Injection.scala:82-103);
    data y = Math.exp(g * x0 + x0).  1 + 1000 gaussians: the first data point consumes the prior draw's cached twin and
    one cached value is left over for the sampler."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = math.exp(rng.next_gaussian() * 1.0 + 0.0)
    ys = np.array([math.exp(rng.next_gaussian() * x0 + x0) for _ in range(1000)])
    g = Graph(1, [0, 1]); z = g.param(0); x = z.exp(); ly = g.col(1, 0)
    u = (ly - x) / x
    row = (models.std_normal_logpdf(u) - x.log()) + ly * -1.0
    return models.ModelSpec("sbc_lognormal", g.compile([models.std_normal_logpdf(z), row]), [np.log(ys)], [0, 1000], 1), rng.r, lambda d: np.exp(d[:, 0])


def exponential_spec():
    """SBCExponential: Exponential(x) = Gamma.standard(1).scale(1/x) (Continuous.scala:152-158): logDensity(y) =
    -(y / (1/x)) - log(1/x).  Data: Marsaglia-Tsang Gamma(1) draws (Continuous.scala:120-146) times 1/x0."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = math.exp(rng.next_gaussian() * 1.0 + 0.0)

    def gamma1():
        a = 1.0
        d = a - 1.0 / 3.0
        c = (1.0 / 3.0) / math.sqrt(d)
        while True:
            xx = rng.next_gaussian(); v = 1.0 + c * xx
            while v <= 0:
                xx = rng.next_gaussian(); v = 1.0 + c * xx
            v3 = v * v * v
            u = rng.next_double()
            if (u < 1 - 0.0331 * xx * xx * xx * xx) or (math.log(u) < 0.5 * xx * xx + d * (1 - v3 + math.log(v3))):
                return d * v3
    ys = np.array([gamma1() * (1.0 / x0) for _ in range(1000)])
    g = Graph(1, [0, 1]); z = g.param(0); x = z.exp(); y = g.col(1, 0)
    inv = 1.0 / x
    row = (y / inv) * -1.0 - inv.log()
    return models.ModelSpec("sbc_exponential", g.compile([models.std_normal_logpdf(z), row]), [ys], [0, 1000], 1), rng.r, lambda d: np.exp(d[:, 0])


@pytest.mark.parametrize("name,builder", [("SBCLogNormal", lognormal_spec), ("SBCExponential", exponential_spec)])
def test_oracle_reproduces_lognormal_prior_goldsets(oracle, name, builder):
    spec, rstate, predict_fn = builder()
    gold = np.array(ALL["models"][name]["goldset"])
    cfg = O.make_config(sampler=O.HMC, n_steps=1, iterations=len(gold), warmup=ALL["warmup"], step_tuner=O.STEP_DUALAVG,
                        delta=0.8, mass_tuner=O.MASS_IDENTITY, math_mode=O.JM_LIBM)
    d = O.OracleDensity(spec, O.JM_LIBM)
    draws, _, st, rc = O.sample_chain_state(d.fn_ptr, d.handle, 1, cfg, rstate)
    rel = np.abs((predict_fn(draws) - gold) / gold)
    assert rc == 0 and rel.max() < 1e-10, (name, rel.max())


# ---- the remaining enabled goldsets of SBCTest.scala:19-34: Gamma, Binomial, Binomial (Poisson regime), NegativeBinomial, LargePoisson ----
def _nemes(z):
    """Combinatorics.gamma -> approxGamma (core/Combinatorics.scala:27-36), on doubles or numpy arrays."""
    v = z + 1
    w = v + (1.0 / ((12 * v) - (1.0 / (10 * v))))
    return (np.log(np.pi * 2) / 2) - (np.log(v) / 2) + (v * (np.log(w) - 1)) - np.log(z)


def _nemes_expr(z):
    v = z + 1.0
    w = v + (1.0 / ((v * 12.0) - (1.0 / (v * 10.0))))
    return (float(np.log(np.pi * 2)) / 2.0 - (v.log() / 2.0)) + (v * (w.log() - 1.0)) - z.log()


def _scaled_uniform_prior(g, lo, hi):
    """Uniform(lo, hi).latent = standard.latent * (hi - lo) + lo (Continuous.scala:205-215, Injection.scala:48-86)."""
    x01, prior = _uniform01_prior(g)
    return x01 * (hi - lo) + lo, prior


def gamma_spec():
    """SBCGamma: Gamma(x, x) = Gamma.standard(x).scale(x), x ~ LogNormal(0,1).  Generator (Continuous.scala:112-146): shape < 1 ->
    u = uniform, then Marsaglia-Tsang at shape + 1, times u^(1/shape); then * scale.  logDensity(y) = standard(y / x) - log x,
    standard(t) = (x - 1) log t - logGamma(x) - t."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = math.exp(rng.next_gaussian() * 1.0 + 0.0)

    def generate(a):
        d = a - 1.0 / 3.0
        c = (1.0 / 3.0) / math.sqrt(d)
        while True:
            xx = rng.next_gaussian(); v = 1.0 + c * xx
            while v <= 0:
                xx = rng.next_gaussian(); v = 1.0 + c * xx
            v3 = v * v * v
            u = rng.next_double()
            if (u < 1 - 0.0331 * xx * xx * xx * xx) or (math.log(u) < 0.5 * xx * xx + d * (1 - v3 + math.log(v3))):
                return d * v3

    def standard(a):
        if a < 1:
            u = rng.next_double()
            return generate(a + 1) * math.pow(u, 1.0 / a)
        return generate(a)
    ys = np.array([standard(x0) * x0 for _ in range(1000)])
    g = Graph(1, [0, 1]); z = g.param(0); x = z.exp(); y = g.col(1, 0)
    t = y / x
    row = (((x - 1.0) * t.log() - _nemes_expr(x)) - t) + x.log() * -1.0
    return models.ModelSpec("sbc_gamma", g.compile([models.std_normal_logpdf(z), row]), [ys], [0, 1000], 1), rng.r, lambda d: np.exp(d[:, 0])


def _binomial_rows(g, x, k, vs):
    """Binomial(p, k).logDensity(v) = Multinomial(Map(true -> p, false -> 1 - p), k).logDensity(Map(true -> v, false -> k - v))
    (Discrete.scala:186-232, Multinomial.scala:17-24): factorial(k) + sum_t (eq(i_t, 0, 0, i_t log p_t) - factorial(i_t))."""
    vs = np.asarray(vs, dtype=np.float64); kv = k - vs
    cols = [vs, kv, _nemes(vs + 1), _nemes(kv + 1)]
    v, w, fv, fw = (g.col(1, j) for j in range(4))
    row = float(_nemes(k + 1.0)) + ((g.eq(v, 0.0, 0.0, v * x.log()) - fv) + (g.eq(w, 0.0, 0.0, w * (1.0 - x).log()) - fw))
    return row, cols


def binomial_spec():
    """SBCBinomial: Binomial(x, 10), x ~ Uniform(0,1).  k < 100 -> the Multinomial generator: k categorical draws, `true` when
    cdf(true) = x0 >= uniform (Generator.scala:129-141)."""
    rng = O.JavaRandom(ALL["seed"]); x0 = rng.next_double()
    vs = [float(sum(1 for _ in range(10) if x0 >= rng.next_double())) for _ in range(1000)]
    g = Graph(1, [0, 4]); x, prior = _uniform01_prior(g)
    row, cols = _binomial_rows(g, x, 10.0, vs)
    return models.ModelSpec("sbc_binomial", g.compile([prior, row]), cols, [0, 1000], 1), rng.r, lambda d: 1 / (1 + np.exp(-d[:, 0]))


def binomial_poisson_spec():
    """SBCBinomialPoissonApproximation: Binomial(x, 200), x ~ Uniform(0, 0.04).  k >= 100 and k p <= 10 -> Poisson(p k) draws
    (Knuth's product method, lambda < 30: Discrete.scala:141-153) capped at k (Discrete.scala:198-204)."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = rng.next_double() * 0.04 + 0.0
    lam = x0 * 200.0

    def small():
        l = math.exp(-lam)
        if l >= 1.0:
            return 0
        k = 0; p = 1.0
        while p > l:
            k += 1; p *= rng.next_double()
        return k - 1
    vs = [float(min(small(), 200)) for _ in range(1000)]
    g = Graph(1, [0, 4]); x, prior = _scaled_uniform_prior(g, 0.0, 0.04)
    row, cols = _binomial_rows(g, x, 200.0, vs)
    return (models.ModelSpec("sbc_binomial_poisson", g.compile([prior, row]), cols, [0, 1000], 1), rng.r,
            lambda d: (1 / (1 + np.exp(-d[:, 0]))) * 0.04 + 0.0)


def negative_binomial_spec():
    """SBCNegativeBinomial: NegativeBinomial(p = x, n = 10), x ~ Uniform(0,1).  Generator: the sum of n Geometric(1 - p) draws
    (Discrete.scala:88-110); logDensity(v) = factorial(n + v - 1) - factorial(v) - factorial(n - 1) + n log(1 - p) + v log p."""
    import math
    rng = O.JavaRandom(ALL["seed"]); x0 = rng.next_double()
    q = 1.0 - x0

    def geometric():
        u = rng.next_double()
        return int(math.floor(math.log(u) / math.log(1 - q)))
    vs = np.array([float(sum(geometric() for _ in range(10))) for _ in range(1000)])
    cols = [vs, _nemes(10.0 + vs - 1 + 1) - _nemes(vs + 1)]
    g = Graph(1, [0, 2]); x, prior = _uniform01_prior(g); v, cf = g.col(1, 0), g.col(1, 1)
    row = ((cf - float(_nemes(10.0 - 1 + 1))) + (1.0 - x).log() * 10.0) + v * x.log()
    return models.ModelSpec("sbc_negbin", g.compile([prior, row]), cols, [0, 1000], 1), rng.r, lambda d: 1 / (1 + np.exp(-d[:, 0]))


def large_poisson_spec():
    """SBCLargePoisson: Poisson(1000 x), x ~ Uniform(0.8, 1).  lambda >= 30 -> the logistic-envelope rejection sampler with the
    rough log-factorial (Discrete.scala:156-189); logDensity(v) = log(lambda) v - lambda - factorial(v)."""
    import math
    rng = O.JavaRandom(ALL["seed"]); width = 1.0 - 0.8
    x0 = rng.next_double() * width + 0.8
    lam = x0 * 1000.0
    c = 0.767 - 3.36 / lam
    beta = math.pi / math.sqrt(3.0 * lam)
    alpha = beta * lam
    kk = math.log(c) - lam - math.log(beta)

    def log_factorial(n):
        xx = float(n + 1)
        return ((xx - 0.5) * math.log(xx)) - xx + (0.5 * math.log(2 * math.pi))

    def large():
        while True:
            u = rng.next_double()
            xx = (alpha - math.log((1.0 - u) / u)) / beta
            n = int(math.floor(xx + 0.5))
            if n >= 0:
                v = rng.next_double()
                yy = alpha - beta * xx
                lhs = yy + math.log(v / math.pow(1.0 + math.exp(yy), 2))
                rhs = kk + n * math.log(lam) - log_factorial(n)
                if lhs <= rhs:
                    return n
    vs = np.array([float(large()) for _ in range(1000)])
    cols = [vs, _nemes(vs + 1)]
    g = Graph(1, [0, 2]); x, prior = _scaled_uniform_prior(g, 0.8, 0.8 + width); v, fv = g.col(1, 0), g.col(1, 1)
    lam_e = x * 1000.0
    row = (lam_e.log() * v - lam_e) - fv
    return (models.ModelSpec("sbc_large_poisson", g.compile([prior, row]), cols, [0, 1000], 1), rng.r,
            lambda d: (1 / (1 + np.exp(-d[:, 0]))) * width + 0.8)


MORE = [("SBCGamma", gamma_spec), ("SBCBinomial", binomial_spec), ("SBCBinomialPoissonApproximation", binomial_poisson_spec),
        ("SBCNegativeBinomial", negative_binomial_spec), ("SBCLargePoisson", large_poisson_spec)]


@pytest.mark.parametrize("name,builder", MORE)
def test_oracle_reproduces_remaining_goldsets(oracle, name, builder):
    spec, rstate, predict_fn = builder()
    gold = np.array(ALL["models"][name]["goldset"])
    cfg = O.make_config(sampler=O.HMC, n_steps=1, iterations=len(gold), warmup=ALL["warmup"], step_tuner=O.STEP_DUALAVG,
                        delta=0.8, mass_tuner=O.MASS_IDENTITY, math_mode=O.JM_LIBM)
    d = O.OracleDensity(spec, O.JM_LIBM)
    draws, _, st, rc = O.sample_chain_state(d.fn_ptr, d.handle, 1, cfg, rstate)
    rel = np.abs((predict_fn(draws) - gold) / gold)
    assert rc == 0 and rel.max() < 1e-10, (name, rel.max(), predict_fn(draws)[:3], gold[:3])
