"""Parity at the BASELINE.json sizes, through the kernels the bench times ("next" #1, row g1).

Every (logp, gradient) here is read back from the tick engine's batched gradient path -- rh_density_eval_ex(...,
RH_ENGINE_TICK): rh_grad_kernel / rh_grad_glm_kernel / rh_grad_gather_kernel followed by rh_tick_kernel's fixed-order
combine -- in the build bench.py uses (fp_contract + factor_outputs [+ K = 8]), and compared with the ORACLE
(oracle/rir.c: the RIR interpreter with DataFunction's sequential row sum, ir/DataFunction.scala:32-84) under the
bound SURVEY 8(d) states:  |delta| <= 1e-11 * sum_rows |term|  per output for 1e6..1e7 rows.

The oracle costs 0.1 s (cfg 2), ~12 s (cfg 4) and ~70 s (cfg 5) per parameter vector at these sizes, so a handful of
distinct vectors is evaluated and TILED over all chains of the configuration: every chain slot of every wavefront
(K-group position, MFMA chain tile, workgroup wave, row split) is compared with the oracle, and chains that were given
the same vector must agree bit for bit (the reductions are fixed-order).
"""
import os

import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi, models
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

TOL_BIG = 1e-11  # 1e6..1e7 rows


def _tile(distinct, chains, seed):
    """chains parameter vectors drawn (with a fixed permutation) from the `distinct` ones; returns (q, index of each)."""
    idx = np.random.default_rng(seed).permutation(np.arange(chains) % len(distinct))
    return np.ascontiguousarray(distinct[idx]), idx


def _check_tiled(spec, model, distinct, chains, seed, tol, label):
    d = O.OracleDensity(spec)
    q, idx = _tile(distinct, chains, seed)
    lp, g = model.density_batch(q, engine=_capi.ENGINE_TICK)
    got = np.concatenate([lp[:, None], g], axis=1)
    worst = 0.0
    for j, qq in enumerate(distinct):
        ref, ab = d.update_both(qq)
        bound = tol * ab + 1e-300
        rows = got[idx == j]
        if len(rows) == 0:          # fewer chains than distinct vectors
            continue
        assert np.all(rows == rows[0]), (label, j, "chains with identical q disagree: the reduction is not fixed-order")
        err = np.abs(rows[0] - ref) / bound
        worst = max(worst, float(err.max()))
        assert np.all(err <= 1.0), (label, j, float(err.max()), int(np.argmax(err)))
    return worst


def test_cfg2_linreg_1e6_rows_1024_chains_bench_build_vs_oracle():
    """cfg 2 exactly as bench.py builds it: 3 covariates x 1e6 rows, 1024 chains, fp_contract + factor_outputs + K = 8
    -> rh_grad_kernel + tick combine, all 1024 chains against the oracle."""
    spec = models.linreg(n=1_000_000, k=3)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True, grad_chains=8)
    assert "#define RH_GRAD_K 8" in m.hip_source and "#define RH_FP_CONTRACT 1" in m.hip_source
    rng = np.random.default_rng(22)
    distinct = np.concatenate([rng.normal(size=(30, 5)) * 0.6,
                               [[np.log(0.7), 0.5, 1.0, -2.0, 0.5]],      # the data-generating point (gradients cancel to ~0)
                               [[0.0, 0.0, 0.0, 0.0, 0.0]]])
    worst = _check_tiled(spec, m, distinct, 1024, 1, TOL_BIG, "cfg2")
    assert worst <= 1.0
    # the strict (JVM-faithful: no FMA, un-factored) build through the same kernel
    ms = R.Model(spec, device=0, grad_chains=4)
    _check_tiled(spec, ms, distinct[:8], 1024, 2, TOL_BIG, "cfg2-strict")
    # ragged chain counts: the last K-group is partly empty
    for chains in (1, 7, 9, 1023):
        _check_tiled(spec, m, distinct[:4], chains, chains, TOL_BIG, "cfg2-%d" % chains)


def test_cfg2_headline_kernel_at_its_benched_size_fused_vs_two_launches_vs_oracle(monkeypatch):
    """The judged bench line's dominant kernel is rh_grad_fused_kernel at 1e6 rows x 1024 chains; the oracle check above goes through
    rh_grad_kernel.  This closes the link at the benched size (VERDICT r5 next #4a): two static-HMC iterations (L = 32, as bench.py
    runs them) through the fused launches are (1) the same chains bit for bit as the two-launch schedule (RH_FUSE=0), all 1024, and
    (2) two of them within 1e-9 of the ORACLE's chains (oracle/sampler.c on the same 1e6 rows; tame dynamics -- a static step, identity
    mass -- so that rounding differences of the row sums do not grow along the trajectory)."""
    from tests.test_gpu_parity import _oracle_cfg
    spec = models.linreg(n=1_000_000, k=3)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True, grad_chains=8)      # bench.py's build
    cfg = R.make_config(2, 0, R.HMCSampler(32), R.StaticStepSize(2e-4), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    seeds = [1000 + c for c in range(1024)]                                                 # bench.py's seeds

    def run():
        s = R.Sampler(m, cfg, seeds); s.warmup(); s.run(2)
        out = (s.draws(), s.timing()["dominant_kernel"], [st.leapfrogSteps for st in s.stats()[0]])
        s.close()
        return out
    fused = run()
    assert fused[1] == "rh_grad_fused_kernel" and all(n == 64 for n in fused[2])
    for c in (0, 1023):                                                                     # the oracle first
        want, _, _ = O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), seeds[c])
        np.testing.assert_allclose(fused[0][c], want, rtol=1e-9, atol=1e-11, err_msg="fused launches at 1e6 x 1024 differ from the ORACLE (chain %d)" % c)
    monkeypatch.setenv("RH_FUSE", "0")
    plain = run()
    monkeypatch.delenv("RH_FUSE")
    assert plain[1] == "rh_grad_kernel"
    assert np.array_equal(fused[0], plain[0]) and fused[2] == plain[2]


@pytest.fixture(scope="module")
def cfg4_spec():
    return models.logistic(n=10_000_000, k=50)


def test_cfg4_logistic_1e7x50_256_chains_mfma_kernel_vs_oracle(cfg4_spec, monkeypatch):
    """cfg 4 at full size (4.08 GB of fp64 columns resident): rh_grad_glm_kernel (fp64 MFMA) for 256 chains against the
    oracle, and against the VALU gradient kernel (RH_GLM_MFMA=0) on 256 DISTINCT parameter vectors."""
    spec = cfg4_spec
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "rh_grad_glm_kernel" in m.hip_source and "#define RH_GLM_TARGET 1" in m.hip_source
    rng = np.random.default_rng(44)
    distinct = rng.normal(size=(3, 51)) * 0.3
    d = O.OracleDensity(spec)
    q, idx = _tile(distinct, 256, 4)
    lp, g = m.density_batch(q, engine=_capi.ENGINE_TICK)
    got = np.concatenate([lp[:, None], g], axis=1)
    abs_min = None
    for j, qq in enumerate(distinct):
        ref, ab = d.update_both(qq)
        abs_min = ab if abs_min is None else np.minimum(abs_min, ab)
        rows = got[idx == j]
        assert np.all(rows == rows[0])
        err = np.abs(rows[0] - ref) / (TOL_BIG * ab + 1e-300)
        assert np.all(err <= 1.0), ("cfg4", j, float(err.max()), int(np.argmax(err)))
    # all 256 chains distinct: MFMA path == VALU path.  The bound uses half of the smallest sum|term| the oracle saw for
    # vectors of the same distribution (computing sum|term| exactly for 256 vectors would take the oracle an hour).
    qd = rng.normal(size=(256, 51)) * 0.3
    lp1, g1 = m.density_batch(qd, engine=_capi.ENGINE_TICK)
    m.close()
    monkeypatch.setenv("RH_GLM_MFMA", "0")
    mv = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    lp2, g2 = mv.density_batch(qd, engine=_capi.ENGINE_TICK)
    mv.close()
    bound = TOL_BIG * 0.5 * abs_min
    assert np.all(np.abs(lp1 - lp2) <= 2 * bound[0])
    assert np.all(np.abs(g1 - g2) <= 2 * bound[None, 1:])


def test_cfg4_sampler_config_nuts_diag_mass_recovers_beta():
    """cfg 4's real sampler configuration -- NUTS(max depth 10) + DiagonalMassMatrixTuner + DualAvg on 51 parameters through
    the MFMA gradient kernel and the tick engine -- recovers the data-generating coefficients with R-hat < 1.05.
    (2e5 rows keep the warm-up, which starts at depth-10 trees, inside a test's time; the kernel is the same.)"""
    n, k = 200_000, 50
    spec = models.logistic(n=n, k=k)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "rh_grad_glm_kernel" in m.hip_source
    cfg = R.make_config(150, 300, R.NUTSSampler(10), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(50, 1.5, 50, 50),
                        engine=_capi.ENGINE_TICK)
    s = R.Sampler(m, cfg, [4000 + c for c in range(256)])
    s.warmup(); s.run(150)
    assert s.timing()["dominant_kernel"] == "rh_grad_glm_kernel"
    draws = s.draws()
    stats, mass = s.stats()
    diag = R.diagnostics(draws)
    rhat = np.array([r for r, _ in diag])
    assert rhat.max() < 1.05, rhat.max()
    assert 0.6 < np.mean([st.meanAcceptProb for st in stats]) < 0.98
    assert not np.allclose(mass, 1.0)                       # the diagonal mass matrix was adapted
    # the data-generating coefficients (models.logistic_data: X first, then beta, from default_rng(seed = 4)); intercept 0
    rng = np.random.default_rng(4); rng.standard_normal((k, n)); beta_true = rng.standard_normal(k)
    post_mean = draws.reshape(-1, k + 1).mean(axis=0)
    # posterior sd ~ 1 / sqrt(n E[x^2] p(1-p)) ~ 0.035; the MLE itself sits ~1 sd from the truth
    assert abs(post_mean[0]) < 0.05 and np.max(np.abs(post_mean[1:] - beta_true)) < 0.2, np.max(np.abs(post_mean[1:] - beta_true))


@pytest.fixture(scope="module")
def cfg5_case():
    """cfg 5 at full size + the oracle's (logp, gradient, sum|term|) at ONE parameter vector (its interpreter evaluates the reference's
    O(rows x G) Lookup semantics: ~70 s), shared by the fast-build and the strict-build test"""
    G, per = 10_000, 100
    spec = models.hier_negbin(G, per)
    distinct = np.random.default_rng(55).normal(size=(4, spec.n_params)) * 0.3
    ref, ab = O.OracleDensity(spec).update_both(distinct[0])
    return spec, distinct, ref, ab


def test_cfg5_hier_negbin_10k_groups_gather_kernel_vs_oracle(cfg5_case):
    """cfg 5 at full size: 10 000 groups x 100 observations, nVars = 10 004, through rh_grad_gather_kernel (group-major
    segmented reduction) + big-mode combine; 1024 chains.  One parameter vector against the oracle, all chains against a numpy
    closed form."""
    G, per, chains = 10_000, 100, 1024
    spec, distinct, ref0, ab0 = cfg5_case
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "rh_grad_gather_kernel" in m.hip_source and "#define RH_BIGN 1" in m.hip_source
    q, idx = _tile(distinct, chains, 5)
    lp, g = m.density_batch(q)                       # gather-mode models always take the tick path
    got = np.concatenate([lp[:, None], g], axis=1)
    for j in range(len(distinct)):
        rows = got[idx == j]
        assert np.all(rows == rows[0])
    # numpy closed form (the check tools/cfg5_probe.py made by hand in round 1), every distinct vector
    v, crow, gid, x0, x1 = spec.columns[1:]
    gi, nf = gid.astype(int), 10.0
    for j, qq in enumerate(distinct):
        mm, s, b0, b1 = qq[:4]; z = qq[4:]
        eta = 10 * mm + np.exp(s) * z[gi] + b0 * x0 + b1 * x1
        p = 1 / (1 + nf * np.exp(-eta))
        ll = crow + nf * np.log(1 - p) + v * np.log(p)
        ref = (-0.5 * mm * mm - models.HALF_LOG_2PI) + (s - np.exp(s)) + (-0.5 * b0 * b0 - models.HALF_LOG_2PI) \
            + (-0.5 * b1 * b1 - models.HALF_LOG_2PI) + np.sum(-0.5 * z * z - models.HALF_LOG_2PI) + ll.sum()
        w = v * (1 - p) - nf * p
        gz = -z + np.exp(s) * np.bincount(gi, weights=w, minlength=G)
        row = got[idx == j][0]
        assert abs(row[0] - ref) <= 1e-12 * np.abs(ll).sum()
        assert np.max(np.abs(row[5:] - gz)) <= 1e-11 * np.max(np.exp(s) * np.bincount(gi, weights=np.abs(w), minlength=G))
        assert abs(row[3] - (-b0 + np.sum(w * x0))) <= 1e-11 * np.sum(np.abs(w * x0))
    # the oracle, one vector
    err = np.abs(got[idx == 0][0] - ref0) / (TOL_BIG * ab0 + 1e-300)
    assert np.all(err <= 1.0), ("cfg5", float(err.max()), int(np.argmax(err)))
    m.close()


def test_cfg5_hier_negbin_10k_groups_strict_build_vs_oracle(cfg5_case):
    """cfg 5 at full size in a STRICT (JVM-faithful: fdlibm exp / log, IEEE division, no FMA, un-factored outputs) build: the same
    gather kernel + big-mode combine, 64 chains, one parameter vector against the oracle at the same bound -- and against the fast
    build's closed-form link only through it."""
    spec, distinct, ref0, ab0 = cfg5_case
    m = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source and "#define RH_BIGN 1" in m.hip_source and "#define RH_FP_CONTRACT 0" in m.hip_source
    eng = m.engines()
    assert eng["tick"], eng["why"]
    q, idx = _tile(distinct[:2], 64, 6)
    lp, g = m.density_batch(q)
    got = np.concatenate([lp[:, None], g], axis=1)
    for j in range(2):
        rows = got[idx == j]
        assert np.all(rows == rows[0])
    err = np.abs(got[idx == 0][0] - ref0) / (TOL_BIG * ab0 + 1e-300)
    assert np.all(err <= 1.0), ("cfg5-strict", float(err.max()), int(np.argmax(err)))
    m.close()
