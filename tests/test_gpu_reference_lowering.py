"""What the reference's own front end emits for streamed models -- derived `gradientColumns` (compute/Target.scala:27-31) --
through rh_model_create on the device: column canonicalisation (csrc/columns.cpp), the fast-mode re-association
(csrc/refactor.cpp), then the same kernels as the hand-derived natural forms.  Parity is always against the oracle's
interpreter evaluating the ORIGINAL program on ALL its original columns."""
import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi, models
from rainier_amd import modeling as M
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _check(spec, model, qs, tol, math_mode=O.JM_LIBM, engine=None):
    """|device - oracle| <= tol * sum_rows |term| per output (1e-12 for N <= 1e4); the worst ratio that occurred is
    appended to gpurun_out/parity_worst.txt (DESIGN 4 quotes the measured figures per model)"""
    d = O.OracleDensity(spec, math_mode)
    lp, g = model.density_batch(qs) if engine is None else model.density_batch(qs, engine=engine)
    worst = 0.0
    for c, q in enumerate(qs):
        ref, ab = d.update_both(q)
        got = np.concatenate([[lp[c]], g[c]])
        both_nan = np.isnan(got) & np.isnan(ref)         # NaN is data (a scale parameter below zero): it must be NaN on both sides
        ratio = np.where(both_nan, 0.0, np.abs(got - ref) / (ab + 1e-300))
        worst = max(worst, float(np.nanmax(ratio)))
        assert np.all((np.abs(got - ref) <= tol * ab + 1e-300) | both_nan), (spec.name, c, float(np.nanmax(ratio)), got, ref)
    try:
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_worst.txt"), "a") as f:
            f.write("%s %s engine=%s worst |delta| / sum|term| = %.3e (bound %.0e)\n" % (
                spec.name, "strict" if "RH_FP_CONTRACT 0" in model.hip_source else "fast", engine, worst, tol))
    except OSError:
        pass
    return worst


def test_logistic_reference_lowering_strict_and_fast():
    k, n = 8, 5000
    spec = models.logistic_reference(n=n, k=k)
    assert len(spec.columns) == 5 * (k + 1)
    qs = np.random.default_rng(11).normal(size=(7, k + 1)) * 0.5
    # strict: the literal arithmetic, on the base columns only (bit-identical relations; y - 1 differs in a signed zero and stays)
    ms = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)
    assert "NCOLS = %d, COL0 = 0" % (k + 2) in ms.hip_source
    _check(spec, ms, qs, 1e-12, O.JM_DET)
    _check(spec, ms, qs, 1e-12, O.JM_DET, engine=_capi.ENGINE_TICK)
    # fast: y and the k covariates; linear predictor on the narrow GLM path or the VALU kernel, scalar part in closed form
    mf = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "NCOLS = %d, COL0 = 0" % (k + 1) in mf.hip_source and "rh_logit_link(s * eta, sp, sg);" in mf.hip_source
    _check(spec, mf, qs, 1e-12)
    _check(spec, mf, qs, 1e-12, engine=_capi.ENGINE_TICK)
    # the same chains as the hand-derived natural form on the same data, to rounding
    nat = R.Model(models.logistic(n=n, k=k, columns=models.logistic_data(n, k)), device=0, fp_contract=True, factor_outputs=True)
    cfg = R.make_config(4, 0, R.HMCSampler(3), R.StaticStepSize(2e-3), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    # parameter order: the reference numbers parameters by creation (a, then b_k), models.logistic uses the same order
    np.testing.assert_allclose(mf.sample(cfg, seeds=range(5)).chains, nat.sample(cfg, seeds=range(5)).chains, rtol=1e-7, atol=1e-9)


def test_cfg4_shape_reference_lowering_255_columns_on_the_mfma_kernel():
    """50 covariates: 255 columns arrive, 51 are uploaded, rh_grad_glm_kernel + closed-form link run -- the reference's negated
    intercept (`a * -1.0` in its Line) as a scaled predictor."""
    k, n, chains = 50, 20000, 48
    spec = models.logistic_reference(n=n, k=k)
    assert len(spec.columns) == 255
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "static constexpr int P = 51, NOTHER = 1, NTHU = 0, NCOLS = 51;" in m.hip_source
    qs = np.random.default_rng(12).normal(size=(3, k + 1)) * 0.25
    _check(spec, m, qs, 1e-12)
    _check(spec, m, qs, 1e-12, engine=_capi.ENGINE_TICK)
    cfg = R.make_config(3, 0, R.HMCSampler(3), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    s = R.Sampler(m, cfg, list(range(chains))); s.warmup(); s.run(3)
    assert s.timing()["dominant_kernel"] == "rh_grad_glm_kernel"
    a = s.draws(); s.close()
    nat = R.Model(models.logistic(n=n, k=k, columns=models.logistic_data(n, k)), device=0, fp_contract=True, factor_outputs=True)
    np.testing.assert_allclose(a, nat.sample(cfg, seeds=range(chains)).chains, rtol=1e-7, atol=1e-9)


def test_linear_regression_reference_lowering_38_columns():
    n, k = 6000, 4
    cols = models.linreg_data(n, k)
    sigma = M.Exponential(1).latent; alpha = M.Normal(0, 1).latent; betas = M.Normal(0, 1).latentVec(k)
    spec = M.Model.observe_vec(cols[0], cols[1:], lambda *u: M.Normal(alpha + M.Real.sum([ui * bi for ui, bi in zip(u, betas)]), sigma),
                               split=False).compile("linreg_ref_4", inline=False)
    assert len(spec.columns) == 38
    qs = np.random.default_rng(13).normal(size=(6, k + 2)) * 0.5
    ms = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)
    assert "NCOLS = 5, COL0 = 0" in ms.hip_source
    _check(spec, ms, qs, 1e-12, O.JM_DET)
    mf = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "NCOLS = 5, COL0 = 0" in mf.hip_source
    _check(spec, mf, qs, 1e-12)
    _check(spec, mf, qs, 1e-12, engine=_capi.ENGINE_TICK)
    # the residual is computed once per row again: the row code is as short as the natural form's (5 basis sums + k + 1 FMAs)
    nat = R.Model(models.linreg(n=n, k=k, columns=cols), device=0, fp_contract=True, factor_outputs=True)
    row = lambda src: src.split("static RH_DEV void row(")[2].split("static RH_DEV void finish(")[0].split("static RH_DEV void row_g(")[0].count("const double n")
    assert row(mf.hip_source) <= row(nat.hip_source) + 4, (row(mf.hip_source), row(nat.hip_source))


def test_more_than_128_columns():
    """The column pointers live in a device table (rh_model_data.cols): a 200-covariate logistic regression (201 columns) runs on
    the MFMA GLM kernel and on the chain engine."""
    k, n, chains = 200, 3000, 20
    spec = models.logistic(n=n, k=k)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    qs = np.random.default_rng(14).normal(size=(3, k + 1)) * 0.1
    _check(spec, m, qs, 1e-12)
    _check(spec, m, qs, 1e-12, engine=_capi.ENGINE_TICK)


def test_canonicalisation_can_be_switched_off(monkeypatch):
    k, n = 4, 2000
    spec = models.logistic_reference(n=n, k=k)
    monkeypatch.setenv("RH_CANON_COLUMNS", "0")
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "NCOLS = 25, COL0 = 0" in m.hip_source
    qs = np.random.default_rng(15).normal(size=(4, k + 1)) * 0.5
    _check(spec, m, qs, 1e-12)


# ---- with Model.observe's 8-way split, i.e. exactly the TargetGroup the JVM would hand over --------------------------------
def _logistic_split(n, k):
    cols = models.logistic_data(n, k)
    a = M.Normal(0, 1).latent; bs = M.Normal(0, 1).latentVec(k)
    m = M.Model.observe_vec(cols[0], cols[1:], lambda *u: M.Bernoulli((a + M.Real.sum([ui * bi for ui, bi in zip(u, bs)])).logistic), split=True)
    return m.compile("logistic_split_%dx%d" % (k, n)), cols


def test_split_logistic_strict_and_fast():
    k, n = 8, 5000
    spec, cols = _logistic_split(n, k)
    assert spec.nrows == [0, 8, 624] and len(spec.columns) > 300
    qs = np.random.default_rng(21).normal(size=(6, k + 1)) * 0.5
    ms = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)         # literal arithmetic; initial chunk unrolled, columns folded
    assert "#define RH_NROWTARGETS 1\n" in ms.hip_source
    _check(spec, ms, qs, 1e-12, O.JM_DET)
    _check(spec, ms, qs, 1e-12, O.JM_DET, engine=_capi.ENGINE_TICK)
    mf = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)   # gradient re-derived, 8 slots rolled back into rows
    assert "#define RH_NROWTARGETS 1\n" in mf.hip_source and "NCOLS = %d, COL0 = 0" % (k + 1) in mf.hip_source
    assert "rh_logit_link(s * eta, sp, sg);" in mf.hip_source
    _check(spec, mf, qs, 1e-12)
    _check(spec, mf, qs, 1e-12, engine=_capi.ENGINE_TICK)
    nat = R.Model(models.logistic(n=n, k=k, columns=cols), device=0, fp_contract=True, factor_outputs=True)
    cfg = R.make_config(4, 0, R.HMCSampler(3), R.StaticStepSize(2e-3), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    np.testing.assert_allclose(mf.sample(cfg, seeds=range(5)).chains, nat.sample(cfg, seeds=range(5)).chains, rtol=1e-7, atol=1e-9)


def test_cfg4_shape_as_the_reference_hands_it_over_1945_columns():
    k, n, chains = 50, 20000, 48
    spec, cols = _logistic_split(n, k)
    assert len(spec.columns) == 1945 and spec.nrows == [0, 8, 2499]
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "#define RH_NROWTARGETS 1\n" in m.hip_source and "static constexpr int P = 51, NOTHER = 1, NTHU = 0, NCOLS = 51;" in m.hip_source
    qs = np.random.default_rng(22).normal(size=(3, k + 1)) * 0.25
    _check(spec, m, qs, 1e-12)
    _check(spec, m, qs, 1e-12, engine=_capi.ENGINE_TICK)
    cfg = R.make_config(3, 0, R.HMCSampler(3), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner(), engine=_capi.ENGINE_TICK)
    s = R.Sampler(m, cfg, list(range(chains))); s.warmup(); s.run(3)
    assert s.timing()["dominant_kernel"] == "rh_grad_glm_kernel"
    a = s.draws(); s.close()
    # the natural form streams the rows in their original order, this one slot-major: the same sums in a different order
    nat = R.Model(models.logistic(n=n, k=k, columns=cols), device=0, fp_contract=True, factor_outputs=True)
    np.testing.assert_allclose(a, nat.sample(cfg, seeds=range(chains)).chains, rtol=1e-7, atol=1e-9)


def test_glmm_poisson2_reference_benchmark_model_on_the_device():
    """bench/stan/GLMMPoisson2.scala (100 sites x 40 years, 146 parameters, two Lookups over index columns) in the reference's
    model text: 493 columns -> 4; density and gradient against the oracle on the original program, and a short adaptive run."""
    import json, os
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glmm_poisson2.json")))
    spec = models.glmm_poisson2_reference(100, 40, data)
    assert len(spec.columns) == 493
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "NCOLS = 4, COL0 = 0" in m.hip_source and "#define RH_NROWTARGETS 1\n" in m.hip_source
    qs = np.random.default_rng(23).normal(size=(4, 146)) * 0.3
    _check(spec, m, qs, 1e-12)
    _check(spec, m, qs, 1e-12, engine=_capi.ENGINE_TICK)
    import time
    t0 = time.time()
    tr = m.sample(R.make_config(10, 20, R.HMCSampler(5)), seeds=range(8))   # dual averaging + windowed diagonal mass, L = 5
    dt = time.time() - t0
    assert tr.chains.shape == (8, 10, 146) and np.all(np.isfinite(tr.chains)) and all(st.leapfrogSteps > 0 for st in tr.stats)
    grads = sum(st.gradientEvaluations for st in tr.stats) / 8
    print("GLMMPoisson2 (reference lowering, generic Lookup path): %.1f s for ~%d gradients per chain, 8 chains" % (dt, grads))


def test_lowdim_gaussmix_reference_benchmark_model_on_the_device():
    """bench/stan/LowDimGaussMix.scala (two-component normal mixture, 1000 observations, 5 parameters) in the reference's model
    text: logSumExp with its Real.gt max, |sigma|; 174 columns -> 2, 992 + 8 observations.  Density and gradient against the
    oracle on the original program (both math modes), and the two modes of the posterior are found."""
    import json, os
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lowdim_gaussmix.json")))
    spec = models.lowdim_gaussmix_reference(data)
    qs = np.random.default_rng(24).normal(size=(6, 5)) * 0.7
    ms = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)
    _check(spec, ms, qs, 1e-12, O.JM_DET)
    mf = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "NCOLS = 2, COL0 = 0" in mf.hip_source and "#define RH_NROWTARGETS 1\n" in mf.hip_source
    _check(spec, mf, qs, 1e-12)
    _check(spec, mf, qs, 1e-12, engine=_capi.ENGINE_TICK)
    tr = mf.sample(R.make_config(60, 200), seeds=range(16))
    mu = np.sort(tr.chains[:, :, [0, 2]], axis=2).mean(axis=1)                    # per chain (mu1, mu2) up to label switching
    # every chain finds the same two components, one on each side of zero (the data are bimodal around -2.7 and +2.9)
    assert np.all(mu[:, 0] < -0.5) and np.all(mu[:, 1] > 0.5) and mu.std(axis=0).max() < 0.1, mu


def test_ark_and_kidiq_reference_benchmark_models_on_the_device():
    """The two reference benchmarks that need no data columns.  KidIQ (bench/stan/KidIQ.scala): the reference inlines the
    2-covariate regression.  ARK (bench/stan/ARK.scala) observes one value at a time, 195 times: 197 data-free targets, whose
    differing constants rh_model_create lifts into the columns of ONE streamed target (csrc/lift.cpp); parity against the oracle
    evaluating the original 197."""
    import json, os
    here = os.path.dirname(os.path.abspath(__file__))
    kid = models.kidiq_reference(json.load(open(os.path.join(here, "golden", "kidiq.json"))))
    assert kid.columns == [] and kid.n_params == 4
    qs = np.random.default_rng(25).normal(size=(8, 4)) * 0.5
    _check(kid, R.Model(kid, device=0, math_mode=_capi.MATH_STRICT), qs, 1e-13, O.JM_DET)
    _check(kid, R.Model(kid, device=0, fp_contract=True, factor_outputs=True), qs, 1e-12)
    ark = models.ark_reference(json.load(open(os.path.join(here, "golden", "ark.json"))))
    assert len(ark.nrows) == 197 and ark.columns == [] and ark.n_params == 7
    m = R.Model(ark, device=0, fp_contract=True, factor_outputs=True)
    assert "#define RH_NROWTARGETS 1\n" in m.hip_source and "#define RH_NTARGETS 3\n" in m.hip_source   # 195 observations = 195 rows
    _check(ark, m, np.random.default_rng(26).normal(size=(8, 7)) * 0.3, 1e-12)
    tr = m.sample(R.make_config(100, 300), seeds=range(16))
    from rainier_amd.sampler import diagnostics
    ch = tr.chains.copy()
    ch[:, :, 1] = np.abs(ch[:, :, 1])                 # sigma = |latent|: the two signs of the latent are the same model
    rhat = max(r for r, _ in diagnostics(ch))
    assert rhat < 1.3, rhat


def test_hierarchical_models_as_the_front_end_hands_them_over_run_in_gather_mode():
    """(a) cfg 5's shape in the reference's own text -- NegBin-logit, eta = a + tau * z(site) + b x, z = Normal(0,1).latentVec(100)
    created last -- through Model.observe's 8-way split: the z prior arrives in the data-free target and is lifted into a row
    target over the group index (csrc/lift.cpp lift_table_priors), the split is rolled back, the gradient re-derived into
    eq(index, k, g, 0) form: gather mode.  (b) 80 schools, one Model.observe per school: the members differ in a parameter, which
    becomes a Lookup over a lifted index column; the eta prior is lifted as well."""
    from rainier_amd import compute as CC
    rng = np.random.default_rng(4)
    K, n = 100, 1500
    a = M.Normal(0, 1).latent; b = M.Normal(0, 1).latent; tau = M.Exponential(1).latent
    zs = M.Normal(0, 1).latentVec(K)
    site = rng.integers(0, K, n).astype(float); x = rng.normal(size=n); ys = rng.poisson(3.0, n).astype(float)
    fn = lambda s, u: M.NegativeBinomial((a + tau * CC.Lookup.apply(s, zs) + b * u).logistic, 5.0)
    spec = M.Model.observe_vec(ys, [site, x], fn, split=True).compile("raw_table_negbin_split", inline=False)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source
    qs = rng.normal(size=(6, spec.n_params)) * 0.3
    _check(spec, m, qs, 1e-12)
    _check(spec, m, qs, 1e-12, engine=_capi.ENGINE_TICK)
    tr = m.sample(R.make_config(20, 40), seeds=range(8))
    assert np.all(np.isfinite(tr.chains))

    J = 80
    y8, sig = rng.normal(size=J) * 5, rng.uniform(5, 15, size=J)
    mu = M.Normal(0, 5).latent; t8 = M.Cauchy(0, 5).latent.abs(); etas = M.Normal(0, 1).latentVec(J)
    mod = M.Model([M.Real.zero])
    for j in range(J):
        mod = M.Model.observe([float(y8[j])], M.Normal(mu + t8 * etas[j], float(sig[j]))).merge(mod)
    schools = mod.compile("schools80")
    qs = rng.normal(size=(6, schools.n_params)) * 0.4
    ms = R.Model(schools, device=0, math_mode=_capi.MATH_STRICT)
    assert "#define RH_HAS_GATHER 1\n" in ms.hip_source and "#define RH_NROWTARGETS 2\n" in ms.hip_source
    _check(schools, ms, qs, 1e-12, O.JM_DET)
    mf = R.Model(schools, device=0, fp_contract=True, factor_outputs=True)
    _check(schools, mf, qs, 1e-12)
    _check(schools, mf, qs, 1e-12, engine=_capi.ENGINE_TICK)
    assert np.all(np.isfinite(mf.sample(R.make_config(20, 40), seeds=range(8)).chains))


@pytest.mark.parametrize("case", ["bernoulli-logit, fast build", "normal regression, JVM-faithful build"])
def test_a_second_data_set_of_the_same_model_does_not_compile_again(case, tmp_path, monkeypatch):
    """Compile once per model shape, on the device: Model.observe of a regression on two data sets (different observations in the
    initial chunk, core/Model.scala:71-132) -- the first create compiles into an EMPTY kernel cache, the second finds every code
    object it needs there: rh_compile_count() does not move, and both models evaluate against the oracle.  The fast build rolls
    the chunk into the streamed rows; the JVM-faithful build keeps it where the front end folded it and reads its observations from
    the model's constant pool (rh_model_data.kpool)."""
    from tests.test_emitter_host import _observe_logistic, _observe_normal
    monkeypatch.setenv("RH_KERNEL_CACHE", str(tmp_path))     # cold: nothing of the in-tree cache is visible
    L = _capi.lib()
    mk, fast = (_observe_logistic, dict(fp_contract=True, factor_outputs=True)) if case.startswith("bernoulli") else (_observe_normal, dict(math_mode=_capi.MATH_STRICT))
    c0 = L.rh_compile_count()
    spec1 = mk(11)
    m1 = R.Model(spec1, device=0, **fast)
    c1 = L.rh_compile_count()
    assert c1 > c0, "the first create must have compiled (the cache directory was empty)"
    spec2 = mk(12)
    assert not np.array_equal(spec1.columns[0][:3], spec2.columns[0][:3]) or not np.array_equal(spec1.columns[1][:3], spec2.columns[1][:3])
    m2 = R.Model(spec2, device=0, **fast)
    assert L.rh_compile_count() == c1, "a second data set of the same model compiled again: the generated source depends on the data"
    assert m1.hip_source == m2.hip_source
    for spec, m in ((spec1, m1), (spec2, m2)):
        q = np.random.default_rng(5).normal(size=(2, spec.n_params)) * 0.3
        lp, g = m.density_batch(q)
        d = O.OracleDensity(spec)
        for c in range(2):
            ref, ab = d.update_both(q[c])
            got = np.concatenate([[lp[c]], g[c]])
            assert np.all(np.abs(got - ref) <= 1e-11 * ab + 1e-300)
    # a few iterations of each (the sampler kernels read the pool too: their data-free targets are inlined copies)
    cfg = R.make_config(3, 3, R.HMCSampler(3), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner())
    for spec, m in ((spec1, m1), (spec2, m2)):
        tr = m.sample(cfg, seeds=[71, 72])
        want, _, _ = O.sample_model(spec, O.make_config(sampler=O.HMC, n_steps=3, iterations=3, warmup=3, step_tuner=O.STEP_STATIC, static_step=1e-3,
                                                        mass_tuner=O.MASS_IDENTITY, math_mode=O.JM_DET), 71)
        np.testing.assert_allclose(tr.chains[0], want, rtol=1e-9, atol=1e-11)
    m1.close(); m2.close()
