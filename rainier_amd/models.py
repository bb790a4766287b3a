"""RIR programs + synthetic data for the BASELINE.json configs.

Each builder returns a `ModelSpec(name, rir, columns, nrows, n_params, meta)`; `columns` is the
flattened list (target order, then column order) of float64 numpy arrays the C ABI expects.

Parameterisations follow the reference's own arithmetic: location-scale families are
non-centred (`Normal(mu, s).latent = z*s + mu`, rainier-core/.../core/Continuous.scala:28-67),
positive supports are log-transformed (core/Support.scala:78-84) and the priors are the standard
densities of Continuous.scala:63-77.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from .frontend import Graph

HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)  # Normal.logDensity, core/Continuous.scala:63-67


@dataclass
class ModelSpec:
    name: str
    rir: bytes
    columns: List[np.ndarray]
    nrows: List[int]           # per target (0 for data-free targets)
    n_params: int
    meta: Dict = field(default_factory=dict)

    @property
    def rows_streamed(self) -> int:
        return int(sum(self.nrows))

    @property
    def bytes_per_row(self) -> int:
        # algorithmic bytes per row-chain eval = 8*(K+1): base columns only
        return 8 * len(self.columns)


def std_normal_logpdf(z):
    return (z * z) / -2.0 - HALF_LOG_2PI


def funnel(dim: int = 10) -> ModelSpec:
    """cfg 1 -- Neal's funnel in Rainier's parameter space (README.md:44):
    y = Normal(0,3).latent = 3 z0, x_i = Normal(0, exp(y/2)).latent = z_i exp(3 z0 / 2).
    Non-centring makes the *sampled* density an isotropic standard normal; the funnel shape only
    appears after predict().  Targets: "prior" + the constant-zero Model.track likelihood
    (core/Model.scala:67)."""
    g = Graph(dim, [0, 0])
    prior = g.sum([std_normal_logpdf(g.param(i)) for i in range(dim)])
    rir = g.compile([prior, g.const(0.0)])
    return ModelSpec("funnel%d" % dim, rir, [], [0, 0], dim, {"kind": "funnel"})


def linreg_data(n: int, k: int = 3, seed: int = 20260925):
    """Synthetic cfg-2 data: X~N(0,1), y = 0.5 + X.(1,-2,0.5) + 0.7 N(0,1)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((k, n))
    beta = np.array([1.0, -2.0, 0.5] + [0.25] * max(0, k - 3))[:k]
    y = 0.5 + beta @ X + 0.7 * rng.standard_normal(n)
    return [np.ascontiguousarray(y)] + [np.ascontiguousarray(X[j]) for j in range(k)]


def linreg(n: int = 1_000_000, k: int = 3, seed: int = 20260925, columns=None) -> ModelSpec:
    """cfg 2 -- README linear regression (README.md:19-32), UN-INLINED.
    theta = (s, a, b_0..b_{k-1}); sigma = exp(s) (Exponential(1).latent: prior s - e^s),
    a, b ~ N(0,1).  Row term: Normal(mu, sigma).logDensity(y) =
    -((y-mu)^2) e^{-2s}/2 - s - 0.5 log 2pi."""
    g = Graph(2 + k, [0, 1 + k])
    s, a = g.param(0), g.param(1)
    b = [g.param(2 + j) for j in range(k)]
    prior = (s - s.exp()) + std_normal_logpdf(a)
    for bj in b:
        prior = prior + std_normal_logpdf(bj)
    y = g.col(1, 0)
    x = [g.col(1, 1 + j) for j in range(k)]
    mu = a
    for bj, xj in zip(b, x):
        mu = mu + bj * xj
    r = y - mu
    inv_var = (s * -2.0).exp()
    row = (r * r) * inv_var / -2.0 - s - HALF_LOG_2PI
    rir = g.compile([prior, row])
    cols = linreg_data(n, k, seed) if columns is None else columns
    return ModelSpec("linreg_%dx%d" % (k, n), rir, cols, [0, n], 2 + k,
                     {"kind": "linreg", "k": k, "flops_per_row": 4 * k + 4})


EIGHT_SCHOOLS_Y = (28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0)       # bench/stan/EightSchools.scala:22
EIGHT_SCHOOLS_SIGMA = (15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0)  # bench/stan/EightSchools.scala:23


def eight_schools() -> ModelSpec:
    """cfg 3 -- bench/stan/EightSchools.scala:9-20.
    theta = (m, c, z_1..z_8): mu = 5m, tau = |5c| (Cauchy(0,5).latent.abs), theta_i = z_i tau + mu.
    Targets: prior, Model.empty's zero likelihood, then 8 single-observation Normal likelihoods
    (each its own data-free target; a 1-element observation never becomes a Column)."""
    n = 10
    g = Graph(n, [0] * 10)
    m, c = g.param(0), g.param(1)
    z = [g.param(2 + i) for i in range(8)]
    cauchy = ((c * c + 1.0) * math.pi).log() * -1.0  # Cauchy.logDensity core/Continuous.scala:72-77
    prior = std_normal_logpdf(m) + cauchy
    for zi in z:
        prior = prior + std_normal_logpdf(zi)
    mu = m * 5.0
    tau = (c * 5.0).abs()
    liks = []
    for zi, y, sg in zip(z, EIGHT_SCHOOLS_Y, EIGHT_SCHOOLS_SIGMA):
        theta = zi * tau + mu
        u = (theta * -1.0 + y) / sg
        liks.append(std_normal_logpdf(u) - math.log(sg))
    rir = g.compile([prior, g.const(0.0)] + liks)
    return ModelSpec("eight_schools", rir, [], [0] * 10, n, {"kind": "eight_schools"})


def logistic_data(n: int, k: int = 50, seed: int = 4):
    """cfg-4 data: X~N(0,1)/sqrt(k), beta~N(0,1), y~Bernoulli(sigmoid(X beta))."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((k, n)) / math.sqrt(k)
    beta = rng.standard_normal(k)
    p = 1.0 / (1.0 + np.exp(-(beta @ X)))
    y = (rng.random(n) < p).astype(np.float64)
    return [np.ascontiguousarray(y)] + [np.ascontiguousarray(X[j]) for j in range(k)]


def logistic(n: int = 10_000_000, k: int = 50, seed: int = 4, columns=None) -> ModelSpec:
    """cfg 4 -- logistic GLM.  theta = (a, b_0..b_{k-1}) ~ N(0,1).
    Bernoulli(p).logDensity(v) = Real.eq(v, 0, log(1-p), log p) (core/Discrete.scala:50-51),
    p = 1/(1+exp(-eta)) written naively like the reference (compute/Real.scala:42)."""
    g = Graph(1 + k, [0, 1 + k])
    a = g.param(0)
    b = [g.param(1 + j) for j in range(k)]
    prior = std_normal_logpdf(a)
    for bj in b:
        prior = prior + std_normal_logpdf(bj)
    y = g.col(1, 0)
    eta = a
    for j, bj in enumerate(b):
        eta = eta + bj * g.col(1, 1 + j)
    p = 1.0 / ((eta * -1.0).exp() + 1.0)
    row = g.eq(y, 0.0, (1.0 - p).log(), p.log())
    rir = g.compile([prior, row])
    cols = logistic_data(n, k, seed) if columns is None else columns
    return ModelSpec("logistic_%dx%d" % (k, n), rir, cols, [0, n], 1 + k,
                     {"kind": "logistic", "k": k, "flops_per_row": 4 * k + 10})


def normal_1d() -> ModelSpec:
    """The fake density of rainier-test/.../sampler/LeapFrogTest.scala:5-13: density -x^2/2, gradient -x."""
    g = Graph(1, [0])
    x = g.param(0)
    rir = g.compile([(x * x) / -2.0])
    return ModelSpec("normal1d", rir, [], [0], 1, {"kind": "normal1d"})


def fit_normal(data=(1.0, 2.0, 3.0)) -> ModelSpec:
    """OptimizerTest's "fit normal" (rainier-test/.../optimizer/OptimizerTest.scala:8-13): mu = Normal(0,10).latent = 10 z,
    sigma = Uniform(0,1).latent = logistic(u) with log-Jacobian log sigma + log(1 - sigma) (core/Support.scala:56-96),
    observations Normal(mu, sigma).  theta = (z, u)."""
    g = Graph(2, [0, 1])
    z, u = g.param(0), g.param(1)
    sigma = 1.0 / ((u * -1.0).exp() + 1.0)
    prior = std_normal_logpdf(z) + sigma.log() + (1.0 - sigma).log()
    y = g.col(1, 0)
    e = (y - z * 10.0) / sigma
    row = (e * e) / -2.0 - sigma.log() - HALF_LOG_2PI
    rir = g.compile([prior, row])
    ys = np.asarray(data, dtype=np.float64)
    return ModelSpec("fit_normal", rir, [ys], [0, len(ys)], 2, {"kind": "fit_normal"})


def funnel_predict(dim: int = 10):
    """Requirements of cfg 1's predict(): y = 3 z0, x_i = z_i exp(y/2) -- where the funnel shape appears."""
    g = Graph(dim, [])
    y = g.param(0) * 3.0
    xs = [g.param(i) * (y / 2.0).exp() for i in range(1, dim)]
    return g.compile_requirements([y] + xs), dim


def nemes_log_gamma(z):
    """Combinatorics.gamma (core/Combinatorics.scala:10-35): log Gamma by Nemes' approximation on z+1 minus log z, with the
    reference's exact special cases gamma(0) = inf, gamma(1) = gamma(2) = 0.  (numpy, for data-only columns.)"""
    z = np.asarray(z, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = z + 1.0
        w = v + 1.0 / ((12.0 * v) - (1.0 / (10.0 * v)))
        out = (np.log(2 * math.pi) / 2.0) - (np.log(v) / 2.0) + (v * (np.log(w) - 1.0)) - np.log(z)
    out = np.where(z == 0.0, np.inf, out)
    return np.where((z == 1.0) | (z == 2.0), 0.0, out)


def hier_negbin_data(groups: int, per_group: int, seed: int = 5, n_fail: float = 10.0):
    """cfg-5 data: alpha_g ~ N(1, 0.5), 2 covariates, NegBin(n = 10) counts; rows sorted by group."""
    rng = np.random.default_rng(seed)
    alpha = 1.0 + 0.5 * rng.standard_normal(groups)
    beta = np.array([0.3, -0.2])
    n = groups * per_group
    gid = np.repeat(np.arange(groups), per_group).astype(np.float64)   # the reference stores indices as doubles
    X = rng.standard_normal((2, n))
    lam = np.exp(alpha[gid.astype(int)] + beta @ X)
    v = rng.negative_binomial(n_fail, n_fail / (n_fail + lam)).astype(np.float64)
    # data-only part of NegativeBinomial.logDensity, folded into a column at build time like the reference does
    crow = nemes_log_gamma(n_fail + v - 1 + 1) - nemes_log_gamma(v + 1) - float(nemes_log_gamma(n_fail - 1 + 1))
    return v, crow, gid, X[0].copy(), X[1].copy()


def hier_negbin(groups: int = 10_000, per_group: int = 100, seed: int = 5, n_fail: float = 10.0) -> ModelSpec:
    """cfg 5 -- hierarchical negative-binomial GLM.  theta = (m, s, b0, b1, z_0..z_{G-1}):
    mu = 10 m (Normal(0,10).latent), sigma_alpha = exp(s) (prior s - e^s), b ~ N(0,1), alpha_g = mu + sigma_alpha z_g.
    The group effect is `alphas(site)` with `site` a Column: RealVec.apply(index: Real) = Lookup(index, reals)
    (compute/Vec.scala:54-59); here the table holds the z parameters and the affine map follows the lookup.
    NegativeBinomial(p, n).logDensity(v) = [data-only factorial terms] + n log(1-p) + v log p (core/Discrete.scala:111-114)
    with p = 1 / (1 + n exp(-eta)) so that the mean n p/(1-p) is exp(eta).  The z prior is written as a row target over
    the G group ids (one row per group) so that it is lowered by the same gather/scatter machinery."""
    G = int(groups)
    n_params = 4 + G
    g = Graph(n_params, [0, 1, 5])
    m, s, b0, b1 = (g.param(i) for i in range(4))
    zs = [g.param(4 + k) for k in range(G)]
    prior = std_normal_logpdf(m) + (s - s.exp()) + std_normal_logpdf(b0) + std_normal_logpdf(b1)
    zprior = std_normal_logpdf(g.lookup(g.col(1, 0), zs, 0))
    v, crow, gid, x0, x1 = (g.col(2, j) for j in range(5))
    alpha = m * 10.0 + s.exp() * g.lookup(gid, zs, 0)
    eta = alpha + b0 * x0 + b1 * x1
    p = 1.0 / ((eta * -1.0).exp() * n_fail + 1.0)
    row = crow + (1.0 - p).log() * n_fail + v * p.log()
    rir = g.compile([prior, zprior, row])
    dv, dc, dg, dx0, dx1 = hier_negbin_data(G, per_group, seed, n_fail)
    cols = [np.arange(G, dtype=np.float64), dv, dc, dg, dx0, dx1]
    return ModelSpec("hier_negbin_%dx%d" % (G, per_group), rir, cols, [0, G, G * per_group], n_params,
                     {"kind": "hier_negbin", "groups": G, "flops_per_row": 30})   # ~30 flop per row-chain eval (+ 1 exp, ~4 log)


def hier_negbin_centred(groups: int = 10_000, per_group: int = 100, seed: int = 5, n_fail: float = 10.0) -> ModelSpec:
    """cfg 5's model in its CENTRED parameterisation: theta = (b0, b1, mu, s, alpha_0 .. alpha_{G-1}) with the group effects themselves
    as parameters, alpha_g ~ Normal(mu, e^s) -- what `Real.parameter { a => Normal(mu, sigma).logDensity(a) }`
    (compute/Real.scala:63-78) gives -- instead of the non-centred alpha_g = mu + e^s z_g.  With 100 observations per group the data
    pin every alpha_g, and it is this form whose posterior is close to a product of well-scaled marginals.  Written as the
    reference would hand it over: the alpha prior sits in the DATA-FREE target (a sum of G terms that all read mu and s; its
    gradient with respect to mu and s runs over all entries), the likelihood reads `alphas(site)` = Lookup(index, alphas) with
    eq(index, k, g, 0) gradients.  The loader lifts the prior into a row target over the group index (csrc/lift.cpp, fast builds)
    and the model runs in gather mode."""
    G = int(groups)
    n_params = 4 + G
    g = Graph(n_params, [0, 5])
    b0, b1, mu, s = (g.param(i) for i in range(4))
    al = [g.param(4 + k) for k in range(G)]
    prior = std_normal_logpdf(b0) + std_normal_logpdf(b1) + std_normal_logpdf(mu * 0.5) + (s - s.exp())
    inv_var = (s * -2.0).exp()
    # sum_k Normal(mu, e^s).logDensity(alpha_k) = sum_k -(alpha_k - mu)^2 / (2 e^{2s})  -  G (s + log sqrt(2 pi)): the parameter-only
    # part merged into one term with the coefficient G, as the reference's Line algebra merges equal terms (compute/LineOps.scala)
    prior = prior + (s + HALF_LOG_2PI) * float(-G)
    for a in al:
        d = a - mu
        prior = prior + (d * d) * inv_var * -0.5
    v, crow, gid, x0, x1 = (g.col(1, j) for j in range(5))
    eta = g.lookup(gid, al, 0) + b0 * x0 + b1 * x1
    p = 1.0 / ((eta * -1.0).exp() * n_fail + 1.0)
    row = crow + (1.0 - p).log() * n_fail + v * p.log()
    rir = g.compile([prior, row])
    dv, dc, dg, dx0, dx1 = hier_negbin_data(G, per_group, seed, n_fail)
    return ModelSpec("hier_negbin_centred_%dx%d" % (G, per_group), rir, [dv, dc, dg, dx0, dx1], [0, G * per_group], n_params,
                     {"kind": "hier_negbin_centred", "groups": G, "flops_per_row": 30})   # the same likelihood row as hier_negbin


def random_walk(T: int = 2000, seed: int = 11, obs_sd: float = 0.5) -> ModelSpec:
    """A local-level state-space model with T + 1 parameters -- theta = (s, x_1 .. x_T): x_1 ~ N(0, 1), x_t ~ N(x_{t-1}, e^s),
    y_t ~ N(x_t, obs_sd), prior s - e^s on the log innovation scale -- written the way a Rainier user writes it with one
    Model.observe per time point: everything is data-free.  The prior ties NEIGHBOURING states, so no part of it is a per-entry
    table prior and nothing streams: the generic path with more than 512 parameters (big mode, RH_BIGTH)."""
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.normal(size=T) * 0.3)
    y = x + rng.normal(size=T) * obs_sd
    g = Graph(T + 1, [0, 0])
    s = g.param(0)
    xs = [g.param(1 + t) for t in range(T)]
    inv_var = (s * -2.0).exp()
    prior = (s - s.exp()) + std_normal_logpdf(xs[0]) + (s + HALF_LOG_2PI) * float(-(T - 1))
    for t in range(1, T):
        d = xs[t] - xs[t - 1]
        prior = prior + (d * d) * inv_var * -0.5
    lik = None
    c = 1.0 / (obs_sd * obs_sd)
    for t in range(T):
        e = xs[t] - float(y[t])
        term = (e * e) * (-0.5 * c)
        lik = term if lik is None else lik + term
    lik = lik + float(-T * (math.log(obs_sd) + HALF_LOG_2PI))
    return ModelSpec("random_walk_%d" % T, g.compile([prior, lik]), [], [0, 0], T + 1, {"kind": "random_walk", "y": y})


def negbin_glm(n: int = 100_000, k: int = 3, seed: int = 7, n_fail: float = 5.0) -> ModelSpec:
    """A negative-binomial GLM without group effects: theta = (a, b_0..b_{k-1}) ~ N(0,1), p = 1 / (1 + n e^{-eta}) so that the
    mean is e^eta, NegativeBinomial(p, n).logDensity(v) written as core/Discrete.scala:111-114 does (the data-only factorial
    terms folded into a column).  The streamed twin of cfg 5's row term: exercises the logit-family closed form on the plain
    batched gradient kernel."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((k, n)) * 0.4
    beta = rng.standard_normal(k) * 0.5
    lam = np.exp(0.3 + beta @ X)
    v = rng.negative_binomial(n_fail, n_fail / (n_fail + lam)).astype(np.float64)
    crow = nemes_log_gamma(n_fail + v - 1 + 1) - nemes_log_gamma(v + 1) - float(nemes_log_gamma(n_fail - 1 + 1))
    g = Graph(1 + k, [0, 2 + k])
    a = g.param(0)
    b = [g.param(1 + j) for j in range(k)]
    prior = std_normal_logpdf(a)
    for bj in b:
        prior = prior + std_normal_logpdf(bj)
    vv, cc = g.col(1, 0), g.col(1, 1)
    eta = a
    for j, bj in enumerate(b):
        eta = eta + bj * g.col(1, 2 + j)
    p = 1.0 / ((eta * -1.0).exp() * n_fail + 1.0)
    row = cc + (1.0 - p).log() * n_fail + vv * p.log()
    rir = g.compile([prior, row])
    cols = [np.ascontiguousarray(v), np.ascontiguousarray(crow)] + [np.ascontiguousarray(X[j]) for j in range(k)]
    return ModelSpec("negbin_glm_%dx%d" % (k, n), rir, cols, [0, n], 1 + k, {"kind": "negbin_glm", "k": k})


# ---- the same configurations written the way the REFERENCE writes them, lowered by its own front end ------------------------
# (rainier_amd/modeling.py + compute.py: rainier-core's distributions and Model.observe, rainier-compute's Real algebra,
#  Gradient, PartialEvaluator.inline and Translator, restated).  The RIR of these builders is what Compiler.compileTargets
#  would hand to the back end; the hand-derived builders above are the natural streamed forms SURVEY 8(d) sizes.
def eight_schools_reference() -> ModelSpec:
    """cfg 3 as rainier-benchmark/.../bench/stan/EightSchools.scala:9-20 writes it.  Every single-observation likelihood is
    inlinable, so all ten targets come out data-free, exactly as in the reference."""
    from . import modeling as M
    mu = M.Normal(0, 5).latent
    tau = M.Cauchy(0, 5).latent.abs()
    thetas = M.Normal(mu, tau).latentVec(len(EIGHT_SCHOOLS_SIGMA))
    m = M.Model([M.Real.zero])                                         # Model.empty
    for theta, y, sg in zip(thetas, EIGHT_SCHOOLS_Y, EIGHT_SCHOOLS_SIGMA):
        m = M.Model.observe([float(y)], M.Normal(theta, sg)).merge(m)   # foldLeft(Model.empty) { observe(...).merge(m) }
    spec = m.compile("eight_schools_reference")
    spec.meta["kind"] = "eight_schools"
    return spec


def ark_reference(data) -> ModelSpec:
    """bench/stan/ARK.scala:9-21 in the reference's model text: an AR(5) series observed ONE value at a time --
    `Model.observe(ys(t), Normal(mu, sigma)).merge(m)` 195 times -- so the TargetGroup has one (inlined, data-free) target per
    observation: 197 targets, which rh_model_create turns into one streamed target of 195 rows (csrc/lift.cpp).
    data = {"ys": 200 floats} (tests/golden/ark.json)."""
    from . import modeling as M
    ys = list(data["ys"])
    alpha = M.Normal(0, 10).latent
    sigma = M.Cauchy(0, 2.5).latent.abs()
    betas = M.Normal(0, 10).latentVec(5)
    m = M.Model([M.Real.zero])                                         # Model.empty
    for t in range(5, len(ys)):
        mu = alpha
        for k in range(1, 6):
            mu = mu + betas[k - 1] * ys[t - k]
        m = M.Model.observe([float(ys[t])], M.Normal(mu, sigma)).merge(m)
    return m.compile("ark_reference")


def kidiq_reference(data, n: int = 400) -> ModelSpec:
    """bench/stan/KidIQ.scala:16-27 in the reference's model text: kid_score ~ Normal(b0 + b1 mom_iq + b2 mom_hs, sigma) through
    Model.observe; 2 covariates + intercept distribute into 10 < 20 terms, so the reference inlines it (no data columns).
    data = tests/golden/kidiq.json."""
    from . import modeling as M
    sigma = M.Cauchy(0, 2.5).latent
    betas = M.Normal(0, 10).latentVec(3)
    ys, iq, hs = data["kidScore"][:n], data["momIQ"][:n], data["momHS"][:n]
    m = M.Model.observe_vec(ys, [iq, hs], lambda u, v: M.Normal(betas[0] + betas[1] * u + betas[2] * v, sigma))
    return m.compile("kidiq_reference_%d" % n)


def funnel_reference(dim: int = 10) -> ModelSpec:
    """cfg 1: y = Normal(0, 3).latent, x_i = Normal(0, exp(y / 2)).latent, tracked (Model.track: the likelihood is Real.zero)."""
    from . import modeling as M
    y = M.Normal(0, 3).latent
    xs = [M.Normal(0, (y / 2).exp()).latent for _ in range(dim - 1)]
    spec = M.Model.track([y] + xs).compile("funnel%d_reference" % dim)
    spec.meta["kind"] = "funnel"
    return spec


def linreg_reference(n: int = 1000, k: int = 3, seed: int = 20260925, columns=None, inline: bool = True, split: bool = True) -> ModelSpec:
    """cfg 2 as the reference's README writes it (README.md:19-32): sigma = Exponential(1).latent, alpha = Normal(0,1).latent,
    betas = Normal(0,1).latentVec(k), Model.observe(ys, Vec.from(xs).map { x => Normal(alpha + x.dot(betas), sigma) }).
    With k <= 3 the reference INLINES the likelihood (15 distributed terms < 20): the spec has no data columns at all."""
    from . import modeling as M
    cols = linreg_data(n, k, seed) if columns is None else columns
    sigma = M.Exponential(1).latent
    alpha = M.Normal(0, 1).latent
    betas = M.Normal(0, 1).latentVec(k)
    m = M.Model.observe_vec(cols[0], cols[1:], lambda *u: M.Normal(alpha + M.Real.sum([ui * bi for ui, bi in zip(u, betas)]), sigma), split=split)
    return m.compile("linreg_reference_%dx%d" % (k, n), inline=inline)


def glmm_poisson2_reference(n_sites: int = 100, n_years: int = 40, data=None) -> ModelSpec:
    """The reference's own group-indexed benchmark, bench/stan/GLMMPoisson2.scala:24-53 (BPA Ch.04: counts by site and year),
    in its model text: `alphas(site)` and `yearBetas(year)` are `Vec.apply` over index COLUMNS = Lookup(Column, reals)
    (compute/Vec.scala:54-59).  data = {"year": 40 floats, "counts": 4000 ints} (tests/golden/glmm_poisson2.json)."""
    from . import modeling as M
    from . import compute as C
    year = list(data["year"])[:n_years]
    counts = [c for i, c in enumerate(data["counts"][:n_sites * 40]) if i % 40 < n_years]
    mu = M.Normal(0, 10).latent
    sd_alpha = M.Uniform(0, 2).latent
    alphas = M.Normal(mu, sd_alpha).latentVec(n_sites)
    sd_year = M.Uniform(0, 1).latent
    betas = M.Normal(0, 10).latentVec(3)
    eps = M.Normal(0, sd_year).latentVec(n_years)
    year_betas = [y * betas[0] + y * y * betas[1] + y * y * y * betas[2] + ep for y, ep in zip(year, eps)]
    ys = [float(y) for s in range(n_sites) for y in range(n_years)]
    ss = [float(s) for s in range(n_sites) for y in range(n_years)]
    m = M.Model.observe_vec(counts, [ys, ss], lambda yr, site: M.Poisson((C.Lookup.apply(yr, year_betas) + C.Lookup.apply(site, alphas)).exp()))
    return m.compile("glmm_poisson2_reference_%dx%d" % (n_sites, n_years))


def lowdim_gaussmix_reference(data) -> ModelSpec:
    """bench/stan/LowDimGaussMix.scala:9-24 in the reference's model text: a two-component normal mixture over 1000 observations,
    `Model.observe(ys, mix)` (8 + 8 x 124 split); logDensity = Real.logSumExp (a max through Real.gt selects, two exps, a log).
    data = {"ys": 1000 floats} (tests/golden/lowdim_gaussmix.json)."""
    from . import modeling as M
    def mu_sigma(): return M.Normal(0, 2).latent, M.Normal(0, 2).latent.abs()
    mu1, sigma1 = mu_sigma()
    mu2, sigma2 = mu_sigma()
    theta = M.Beta(5, 5).latent
    mix = M.Mixture([(M.Normal(mu1, sigma1), theta), (M.Normal(mu2, sigma2), M.Real.one - theta)])
    return M.Model.observe(list(data["ys"]), mix).compile("lowdim_gaussmix_reference")


def logistic_reference(n: int = 1000, k: int = 8, seed: int = 4, columns=None) -> ModelSpec:
    """cfg 4's model text: Bernoulli((a + x.dot(b)).logistic) observed row by row.  Not inlinable (the link is non-linear); the
    reference's algebra pushes every data-only factor into derived columns (gradientColumns): 5 (k + 1) columns."""
    from . import modeling as M
    cols = logistic_data(n, k, seed) if columns is None else columns
    a = M.Normal(0, 1).latent
    bs = M.Normal(0, 1).latentVec(k)
    m = M.Model.observe_vec(cols[0], cols[1:], lambda *u: M.Bernoulli((a + M.Real.sum([ui * bi for ui, bi in zip(u, bs)])).logistic), split=False)
    return m.compile("logistic_reference_%dx%d" % (k, n))
