"""Pins for the CPU oracle (oracle/): known answers, the reference's own LeapFrogTest, gradients."""
import ctypes as C
import math

import numpy as np
import pytest

from rainier_amd import models
from tests import oracle_lib as O


def test_java_util_random_known_answers(oracle):
    # published JDK values (java.util.Random Javadoc algorithm); SURVEY.md §8(c)
    assert O.JavaRandom(42).next_int() == -1170105035
    assert O.JavaRandom(0).next_double() == 0.730967787376657
    assert O.JavaRandom(0).next_gaussian() == 0.8025330637390305
    assert O.JavaRandom(42).next_gaussian() == 1.1419053154730547
    r = O.JavaRandom(123)  # ScalaRNG(123L), the seed of LeapFrogTest.scala:16
    got = [r.next_gaussian() for _ in range(3)]
    assert got == [-1.4380493091409068, 0.6341950751776804, 0.22606201283216426]


def test_gaussian_cache_survives_uniform(oracle):
    # nextGaussian caches its 2nd value; an intervening nextDouble must not drop it
    a = O.JavaRandom(7); g1 = a.next_gaussian(); u = a.next_double(); g2 = a.next_gaussian()
    b = O.JavaRandom(7); h1 = b.next_gaussian(); h2 = b.next_gaussian()
    assert g1 == h1 and g2 == h2 and 0.0 <= u < 1.0


def test_fdlibm_log_exp_within_one_ulp(oracle):
    rng = np.random.default_rng(0)
    xs = np.exp(rng.uniform(-700, 700, 20000))
    for x in xs[:5000]:
        a, b = oracle.jm_strict_log(x), math.log(x)
        assert abs(a - b) <= abs(np.spacing(b))
    for y in rng.uniform(-700, 700, 5000):
        a, b = oracle.jm_strict_exp(y), math.exp(y)
        assert abs(a - b) <= abs(np.spacing(b))
    assert oracle.jm_strict_log(0.0) == -math.inf and math.isnan(oracle.jm_strict_log(-1.0))
    assert oracle.jm_strict_exp(-math.inf) == 0.0 and oracle.jm_strict_exp(1.0) == 2.7182818284590455
    for t in (1.0, 2.0, 17.0, 1000.0):
        assert abs(oracle.jm_pow_neg075(O.JM_DET, t) - t ** -0.75) <= 2 * np.spacing(t ** -0.75)


def _leapfrog_test_run(oracle, rng, mass):
    """rainier-test/.../sampler/LeapFrogTest.scala:18-36 `run`, over the RIR normal_1d model."""
    spec = models.normal_1d()
    d = O.OracleDensity(spec)
    lf = oracle.orc_lf_new(d.fn_ptr, d.handle, 1, C.byref(rng), O.JM_LIBM)
    params = np.zeros(3)
    m = None if mass is None else np.array(mass, dtype=np.float64)
    mp = None if m is None else O._dp(m)
    oracle.orc_lf_initialize(lf, mp, O._dp(params))
    out = []
    for _ in range(1000):
        oracle.orc_lf_start_iteration(lf, O._dp(params), mp)
        oracle.orc_lf_take_steps(lf, 1, 1.0, mp)
        oracle.orc_lf_finish_iteration(lf, O._dp(params), mp)
        out.append(params[1])
    oracle.orc_lf_free(lf)
    return np.array(out)


def test_reference_leapfrogtest(oracle):
    # LeapFrogTest.scala:15-78.  `implicit val rng = new ScalaRNG(123L)` is ONE class-level stream:
    # the identity-mass test runs first and the diagonal-mass test continues the same stream.
    # thresholds: |mean| < 0.2, |var(about 0) - 1| < 0.2 (identity) / 0.3 (DiagonalMassMatrix(Array(0.1)))
    rng = O.JRandom(); oracle.jrandom_init(C.byref(rng), 123)
    for mass, eps_var in [(None, 0.2), ([0.1], 0.3)]:
        xs = _leapfrog_test_run(oracle, rng, mass)
        mean = xs.sum() / xs.size
        var = ((xs - 0.0) ** 2).sum() / (xs.size - 1)
        assert abs(mean) < 0.2, (mass, mean)
        assert abs(var - 1.0) < eps_var, (mass, var)


@pytest.mark.parametrize("builder", [
    lambda: models.funnel(10), lambda: models.eight_schools(),
    lambda: models.linreg(n=257, k=3), lambda: models.logistic(n=101, k=5)])
def test_gradient_matches_central_difference(oracle, builder):
    # compute/RealTest.scala:39-52: symbolic gradient == central difference, dx = 1e-5, rel 1e-3
    spec = builder()
    d = O.OracleDensity(spec)
    rng = np.random.default_rng(1)
    for _ in range(3):
        q = rng.normal(size=spec.n_params) * 0.5
        out = d.update(q)
        for i in range(spec.n_params):
            dq = np.zeros_like(q); dq[i] = 1e-5
            fd = (d.update(q + dq)[0] - d.update(q - dq)[0]) / 2e-5
            assert fd == pytest.approx(out[1 + i], rel=1e-3, abs=1e-6)


def test_eight_schools_closed_form(oracle):
    # independent closed form of SURVEY §3.4.3 on the reference's data (EightSchools.scala:22-23)
    spec = models.eight_schools()
    d = O.OracleDensity(spec)
    q = np.array([0.3, -0.7, 0.1, 0.2, -0.3, 0.4, -0.5, 0.6, -0.7, 0.8])
    m, c, z = q[0], q[1], q[2:]
    ys = np.array(models.EIGHT_SCHOOLS_Y); sg = np.array(models.EIGHT_SCHOOLS_SIGMA)
    theta = z * abs(5 * c) + 5 * m
    lp = (-0.5 * m * m - models.HALF_LOG_2PI) - math.log(math.pi * (1 + c * c)) + np.sum(-0.5 * z * z - models.HALF_LOG_2PI)
    lp += np.sum(-0.5 * ((ys - theta) / sg) ** 2 - np.log(sg) - models.HALF_LOG_2PI)
    out = d.update(q)
    assert out[0] == pytest.approx(lp, rel=1e-13)
    resid = (ys - theta) / sg ** 2
    assert out[1] == pytest.approx(-m + 5 * resid.sum(), rel=1e-12)
    assert out[2] == pytest.approx(-2 * c / (1 + c * c) + np.sum(resid * z) * 5 * np.sign(c), rel=1e-12)
    np.testing.assert_allclose(out[3:], -z + resid * abs(5 * c), rtol=1e-12)


def test_linreg_closed_form_and_ld(oracle):
    spec = models.linreg(n=5000, k=3, seed=3)
    d = O.OracleDensity(spec)
    q = np.array([-0.2, 0.4, 0.9, -1.8, 0.45])
    y, X = spec.columns[0], np.stack(spec.columns[1:])
    s, a, b = q[0], q[1], q[2:]
    r = y - a - b @ X
    iv = math.exp(-2 * s)
    lp = (s - math.exp(s)) + np.sum(-0.5 * q[1:] ** 2 - models.HALF_LOG_2PI) + np.sum(-0.5 * r * r * iv - s - models.HALF_LOG_2PI)
    g = np.concatenate([[1 - math.exp(s) + np.sum(r * r * iv - 1)], [-a + iv * r.sum()], -b + iv * (X @ r)])
    out = d.update(q)
    tol = 1e-12 * d.abs_sums(q)
    assert abs(out[0] - lp) <= tol[0]
    assert np.all(np.abs(out[1:] - g) <= tol[1:] + 1e-12)
    assert np.all(np.abs(d.update_ld(q) - out) <= tol)


def test_lookup_out_of_range_is_an_error(oracle):
    from rainier_amd.frontend import Graph
    g = Graph(1, [0])
    x = g.param(0)
    e = g.lookup(x, [g.const(1.0), x * 2.0], low=0)
    spec = models.ModelSpec("lk", g.compile([e]), [], [0], 1)
    d = O.OracleDensity(spec)
    assert d.update(np.array([1.5]))[0] == 3.0      # D2I truncation: 1.5 -> 1
    assert d.update(np.array([0.99]))[0] == 1.0
    with pytest.raises(RuntimeError):
        d.update(np.array([2.0]))                    # generated NPE in the reference (MethodGenerator.scala:164-167)
    with pytest.raises(RuntimeError):
        d.update(np.array([-1.0]))


def test_compare_and_pow_semantics(oracle):
    from rainier_amd.frontend import Graph
    g = Graph(2, [0])
    a, b = g.param(0), g.param(1)
    spec = models.ModelSpec("cp", g.compile([a.compare(b) + (a ** b) * 0.0 + 0.0], gradients=[[g.const(0.0)] * 2]), [], [0], 2)
    d = O.OracleDensity(spec)
    assert d.update(np.array([2.0, 1.0]))[0] == 1.0
    assert d.update(np.array([1.0, 1.0]))[0] == 0.0
    assert d.update(np.array([0.5, 1.0]))[0] == -1.0
    g2 = Graph(2, [0]); a, b = g2.param(0), g2.param(1)
    spec2 = models.ModelSpec("c2", g2.compile([a.compare(b)], gradients=[[g2.const(0.0)] * 2]), [], [0], 2)
    d2 = O.OracleDensity(spec2)
    assert d2.update(np.array([math.nan, 1.0]))[0] == -1.0   # DCMPL: NaN -> -1


def test_trace_diagnostics_matches_numpy(oracle):
    rng = np.random.default_rng(5)
    m, n = 4, 500
    x = np.zeros((m, n))
    for c in range(m):
        for i in range(1, n):
            x[c, i] = 0.7 * x[c, i - 1] + rng.normal()
    rhat, ess = O.diagnostics(x)
    means = x.mean(axis=1); b = n / (m - 1) * ((means - means.mean()) ** 2).sum()
    w = (((x - means[:, None]) ** 2).sum(axis=1) / (n - 1)).mean()
    v = (n - 1) / n * w + b / n
    acc, lag = 0.0, 1
    while True:
        vt = np.mean([((x[c, lag:] - x[c, :-lag]) ** 2).sum() / (n - lag) for c in range(m)])
        pt = 1 - vt / (2 * v)
        if pt > 0 and lag < 100:
            acc += pt; lag += 1
        else:
            break
    assert rhat == pytest.approx(math.sqrt(v / w), rel=1e-12)
    assert ess == pytest.approx(n * m / (1 + 2 * acc), rel=1e-10)
    assert 100 < ess < 1000


def test_driver_runs_and_recovers_posterior(oracle):
    # funnel in parameter space is N(0, I): posterior mean ~0, var ~1 (statistical, 4-sigma bounds)
    spec = models.funnel(10)
    cfg = O.make_config(sampler=O.HMC, n_steps=5, iterations=4000, warmup=1000)
    draws, mass, st = O.sample_model(spec, cfg, 123)
    assert st.leapfrog_steps == 4000 * 5
    assert st.gradient_evaluations == 4000 * 11          # reference evaluates 2L+1 per trajectory
    assert np.all(np.abs(draws.mean(axis=0)) < 0.15)
    assert np.all(np.abs(draws.var(axis=0) - 1.0) < 0.25)
    assert 0.6 < st.mean_accept_prob < 0.95
    # Stats.bfmi (S/Stats.scala:14-16): for a d-dimensional Gaussian with full momentum refresh E-BFMI is ~1 (>= 0.3 is healthy)
    assert 0.5 < st.bfmi < 1.6
    # EHMC + windowed diagonal mass adaptation (DefaultConfig, S/Sampler.scala:17-27)
    cfg = O.make_config(sampler=O.EHMC, max_steps=1024, iterations=2000, warmup=1000, mass_tuner=O.MASS_DIAG_WINDOWED)
    draws, mass, st = O.sample_model(models.eight_schools(), cfg, 7)
    assert np.all(np.isfinite(draws)) and np.all(mass > 0) and not np.allclose(mass, 1.0)
    assert abs(draws[:, 2:].mean()) < 0.3


def test_nuts_extension_recovers_standard_normal(oracle):
    # NUTS is NOT in the reference: statistical sanity of the oracle's statement of the algorithm
    cfg = O.make_config(sampler=O.NUTS, iterations=3000, warmup=800, mass_tuner=O.MASS_DIAG_WINDOWED, nuts_max_depth=10,
                        math_mode=O.JM_DET)
    draws, mass, st = O.sample_model(models.funnel(10), cfg, 11)
    assert np.all(np.abs(draws.mean(axis=0)) < 0.12) and np.all(np.abs(draws.var(axis=0) - 1.0) < 0.2)
    assert 0.7 < st.mean_accept_prob < 0.9 and 1 <= st.leapfrog_steps / 3000 <= 2 ** 10


def test_dense_mass_matrix_pieces(oracle):
    # S/MassMatrix.scala:33-117: U^T U = M, the packed triangular solve, and CholeskyTest's worked example
    rng = np.random.default_rng(0); n = 7
    A = rng.normal(size=(n, n)); M = A @ A.T + n * np.eye(n)
    U = np.zeros(n * (n + 1) // 2)
    oracle.orc_cholesky_upper(O._dp(np.ascontiguousarray(M.ravel())), n, O._dp(U))
    Um = np.zeros((n, n)); l = 0
    for i in range(n):
        for k in range(n - i):
            Um[i, i + k] = U[l]; l += 1
    np.testing.assert_allclose(Um.T @ Um, M, rtol=1e-12)
    b = rng.normal(size=n); x = np.zeros(n)
    oracle.orc_upper_triangular_solve(O._dp(U), O._dp(b), n, O._dp(x))
    np.testing.assert_allclose(Um @ x, b, rtol=1e-12)
    pk = np.array([1, 2, 4, 7, 3, 5, 8, 6, 9, 10], dtype=float); y = np.array([45, 53, 54, 40], dtype=float); x = np.zeros(4)
    oracle.orc_upper_triangular_solve(O._dp(pk), O._dp(y), 4, O._dp(x))     # compute/CholeskyTest.scala:50-81
    assert list(x) == [1.0, 2.0, 3.0, 4.0]
    cfg = O.make_config(sampler=O.EHMC, iterations=800, warmup=600, mass_tuner=O.MASS_DENSE_WINDOWED, math_mode=O.JM_DET)
    dense = np.zeros(100); cfg.dense_out = O._dp(dense)
    draws, mass, st = O.sample_model(models.eight_schools(), cfg, 3)
    assert np.linalg.eigvalsh(dense.reshape(10, 10)).min() > 0 and np.allclose(np.diag(dense.reshape(10, 10)), mass)
    assert np.all(np.isfinite(draws)) and 0.6 < st.mean_accept_prob < 0.95


def test_lbfgs_fit_normal_and_termination(oracle):
    # optimizer/OptimizerTest.scala:8-13 "fit normal"; Optimizer.lbfgs (optimizer/Optimizer.scala:6-24): m = 5, eps = 0.1
    import scipy.optimize as so
    spec = models.fit_normal()
    x, evals = O.optimize_model(spec)
    d = O.OracleDensity(spec)
    out = d.update(x)
    assert evals == 12                                                       # regression pin of the restatement
    assert np.linalg.norm(out[1:]) / max(1.0, np.linalg.norm(x)) <= 0.1      # LBFGS.java:171-175
    ref = so.minimize(lambda q: -d.update(q)[0], np.zeros(2), jac=lambda q: -d.update(q)[1:], method="L-BFGS-B", tol=1e-12)
    assert np.allclose(x, ref.x, atol=0.02)                                  # eps = 0.1 stops this close to the MAP
    mu, sigma = 10 * x[0], 1 / (1 + np.exp(-x[1]))
    assert abs(mu - 2.0) < 0.02 and abs(sigma - 0.69) < 0.01                 # data (1,2,3): mean 2; sigma < 1 by the prior
    # a zero gradient at the start is the reference's RuntimeException("dginit") (LBFGS.java:236-237)
    assert O.optimize_model(models.funnel())[1] == -1
    # max_evals guard
    assert O.optimize_model(spec, max_evals=3)[1] == -2


def test_lbfgs_reverse_communication_on_quadratic(oracle):
    # LBFGS.apply as the reference test drives it (OptimizerTest.scala:32-41): f, g in, x updated in place
    rng = np.random.default_rng(5); n = 12
    A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n); b = rng.normal(size=n)
    lb = O.Lbfgs(n, m=5, eps=1e-7)
    for it in range(500):
        f, g = 0.5 * lb.x @ A @ lb.x - b @ lb.x, A @ lb.x - b
        r = lb.apply(f, g)
        assert r >= 0
        if r == 1:
            break
    assert r == 1 and it < 200
    np.testing.assert_allclose(lb.x, np.linalg.solve(A, b), atol=1e-7)


def test_closed_form_speed_build_matches_interpreter(oracle):
    # oracle/closed_form.c (bench.py's second CPU figure) against the RIR interpreter on cfg 2's model
    spec = models.linreg(n=5000)
    th = np.array([-0.3, 0.5, 1.0, -2.0, 0.5]); out = np.zeros(6); c = spec.columns
    oracle.orc_linreg_streamed(O._dp(c[0]), O._dp(c[1]), O._dp(c[2]), O._dp(c[3]), len(c[0]), O._dp(th), O._dp(out))
    np.testing.assert_allclose(out, O.OracleDensity(spec).update(th), rtol=1e-11)


def test_nuts_checkpoint_bookkeeping_equals_the_recursive_tree(oracle):
    """f2 cross-validation.  NUTS does not exist in the reference, so the iterative multinomial NUTS
    of oracle/sampler.c:nuts_iteration (and with it the device automaton, which is compared bit for bit with it on the
    GPU) is pinned against a SECOND, independent statement of the transition: nuts_iteration_recursive builds the same
    tree the Hoffman-Gelman / Stan way (recursive halves, node = (first r, last r, sum r), no checkpoints, no leaf-index
    arithmetic).  Same uniforms, same leapfrog states => the same leaf must be selected and the trajectory must stop at
    the same leaf: whole chains identical, leapfrog counts identical, over > 1e4 trajectories that include divergences,
    max-depth saturation (depths 1..12), identity / adapted diagonal / static mass."""
    NUTS_REC = 3
    total = 0
    cases = [
        (models.funnel(10), dict(iterations=1500, warmup=300, nuts_max_depth=10), [1, 2]),
        (models.eight_schools(), dict(iterations=1500, warmup=400, mass_tuner=O.MASS_DIAG_WINDOWED, nuts_max_depth=10), [3, 4]),
        (models.normal_1d(), dict(iterations=800, warmup=100, nuts_max_depth=12), [5]),
        # saturation: a tiny static step never turns -> every trajectory runs to max depth (2^d - 1 leaves)
        (models.normal_1d(), dict(iterations=12, warmup=0, step_tuner=O.STEP_STATIC, static_step=1e-5, nuts_max_depth=12), [6]),
        (models.funnel(10), dict(iterations=60, warmup=0, step_tuner=O.STEP_STATIC, static_step=1e-4, nuts_max_depth=7), [7]),
        (models.eight_schools(), dict(iterations=300, warmup=0, step_tuner=O.STEP_STATIC, static_step=0.02, nuts_max_depth=1), [8]),
        (models.eight_schools(), dict(iterations=300, warmup=0, step_tuner=O.STEP_STATIC, static_step=0.05, nuts_max_depth=3), [9]),
        # divergences: a huge static step on the heavy-tailed eight-schools posterior / the 1-d normal
        (models.eight_schools(), dict(iterations=600, warmup=0, step_tuner=O.STEP_STATIC, static_step=6.0, nuts_max_depth=10), [10, 11]),
        (models.normal_1d(), dict(iterations=600, warmup=0, step_tuner=O.STEP_STATIC, static_step=60.0, nuts_max_depth=10), [12]),
        (models.eight_schools(), dict(iterations=500, warmup=0, step_tuner=O.STEP_STATIC, static_step=0.3, nuts_max_depth=10,
                                      mass_tuner=O.MASS_STATIC_DIAG, static_mass=np.linspace(0.3, 3.0, 10)), [13]),
        (models.linreg(n=200, k=3), dict(iterations=600, warmup=300, mass_tuner=O.MASS_DIAG_WINDOWED, nuts_max_depth=10), [14]),
    ]
    depth_seen = set()
    for spec, kw, seeds in cases:
        for seed in seeds:
            a = O.make_config(sampler=O.NUTS, math_mode=O.JM_DET, **kw)
            b = O.make_config(sampler=NUTS_REC, math_mode=O.JM_DET, **kw)
            da, ma, sa = O.sample_model(spec, a, seed)
            db, mb, sb = O.sample_model(spec, b, seed)
            assert np.array_equal(da, db, equal_nan=True), (spec.name, kw, seed)
            assert np.array_equal(ma, mb)
            assert sa.leapfrog_steps == sb.leapfrog_steps and sa.warmup_leapfrog_steps == sb.warmup_leapfrog_steps
            assert sa.mean_accept_prob == sb.mean_accept_prob and sa.step_size == sb.step_size
            total += kw["iterations"] + kw["warmup"]
            if kw["iterations"]:
                depth_seen.add(round(np.log2(sa.leapfrog_steps / kw["iterations"] + 1), 1))
    assert total >= 10000
    # the saturation cases really ran full trees (2^12 - 1 and 2^7 - 1 leaves per iteration) ...
    a = O.make_config(sampler=O.NUTS, math_mode=O.JM_DET, iterations=12, warmup=0, step_tuner=O.STEP_STATIC, static_step=1e-5, nuts_max_depth=12)
    _, _, st = O.sample_model(models.normal_1d(), a, 6)
    assert st.leapfrog_steps == 12 * (2 ** 12 - 1)
    # ... and the divergent ones really diverged early (mean acceptance far below the target, short trees)
    a = O.make_config(sampler=O.NUTS, math_mode=O.JM_DET, iterations=600, warmup=0, step_tuner=O.STEP_STATIC, static_step=60.0, nuts_max_depth=10)
    _, _, st = O.sample_model(models.normal_1d(), a, 12)
    assert st.mean_accept_prob < 0.2 and st.leapfrog_steps < 600 * 8
