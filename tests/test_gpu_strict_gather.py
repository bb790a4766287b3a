"""Strict (JVM-faithful) builds of hierarchical models on the device: the reference's per-entry mask columns read as the selects
they are (csrc/columns.cpp index masks, csrc/rollstrict.cpp; RH_INDEX_MASKS, on by default since round 5 -- DESIGN 8.10).

What it checks: the reference-text cfg-5 shapes (raw table of trailing parameters, NegBin-logit and Poisson-log, with and without
Model.observe's 8-way split, with entries no row selects) in a STRICT build run through rh_grad_gather_kernel -- density and
gradient against the oracle on the ORIGINAL program (compute/Gradient.scala:146-152's eq(index, k, g, 0) gradient, one mask column
per entry) at 1e-12 * sum|term| on both seams, and a short static-HMC run chain for chain against the oracle's sampler;
bench/stan/GLMMPoisson2.scala in a strict build (4 streamed columns instead of 452) the same way."""
import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi
from tests import oracle_lib as O
from tests.test_emitter_host import _raw_table_spec

pytestmark = pytest.mark.gpu

STRICT = dict(math_mode=_capi.MATH_STRICT)


@pytest.mark.parametrize("family,n", [("negbin", 1500), ("negbin-split", 1500), ("poisson", 1500), ("poisson-split", 1500), ("negbin", 150), ("negbin-split", 333)])
def test_strict_reference_text_hierarchical_models_in_gather_mode(family, n):
    spec, qs = _raw_table_spec(family, n=n)
    d = O.OracleDensity(spec)
    refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]      # oracle first
    m = R.Model(spec, device=0, **STRICT)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source
    eng = m.engines()
    assert eng["tick"], eng["why"]
    for splits in (0, 3):
        lp, g = m.density_batch(np.asarray(qs), engine=_capi.ENGINE_TICK, grad_splits=splits)
        for c, (ref, ab) in enumerate(refs):
            got = np.concatenate([[lp[c]], g[c]])
            assert np.all(np.abs(got - ref) <= 1e-12 * ab + 1e-300), (family, splits, c, float(np.max(np.abs(got - ref) / (ab + 1e-300))))
    from tests.test_gpu_parity import _oracle_cfg
    cfg = R.make_config(4, 0, R.HMCSampler(3), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner())
    seeds = [5100, 5101]
    want = [O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), sd)[0] for sd in seeds]
    got = m.sample(cfg, seeds=seeds).chains
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
    m.close()


def test_strict_cfg5_shape_beyond_the_generic_path_s_parameter_limit():
    """600 groups = 603 parameters: outside gather mode a strict build of this model takes the memory-resident generic path"""
    spec, qs = _raw_table_spec("negbin", K=600, n=6000, seed=11)
    d = O.OracleDensity(spec)
    refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]
    m = R.Model(spec, device=0, **STRICT)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source
    lp, g = m.density_batch(np.asarray(qs), engine=_capi.ENGINE_TICK)
    for c, (ref, ab) in enumerate(refs):
        got = np.concatenate([[lp[c]], g[c]])
        assert np.all(np.abs(got - ref) <= 1e-12 * ab + 1e-300)
    m.close()


def test_strict_glmm_poisson2_on_the_device():
    """bench/stan/GLMMPoisson2.scala, strict build: 452 -> 4 streamed columns (generic Lookup path: neither table is a trailing run)"""
    import json, os
    from rainier_amd import models
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glmm_poisson2.json")))
    spec = models.glmm_poisson2_reference(100, 40, data)
    qs = np.random.default_rng(23).normal(size=(3, 146)) * 0.3
    d = O.OracleDensity(spec)
    refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]
    from tests.test_gpu_parity import _oracle_cfg
    cfg = R.make_config(4, 0, R.HMCSampler(3), R.StaticStepSize(1e-3), R.IdentityMassMatrixTuner())
    seeds = [5200, 5201]
    want = [O.sample_model(spec, _oracle_cfg(cfg, O.JM_DET), sd)[0] for sd in seeds]      # oracle first
    import time
    t0 = time.perf_counter()
    m = R.Model(spec, device=0, **STRICT)
    create_s = time.perf_counter() - t0
    assert "NCOLS = 4, COL0 = 0" in m.hip_source
    eng = m.engines()
    assert eng["tick"] and eng["density"], eng["why"]          # every kernel the model launches is fit to run
    assert eng["compile_attempts"] <= 2, eng
    print("strict GLMMPoisson2: rh_model_create %.1f s, %d attempt(s)" % (create_s, eng["compile_attempts"]))
    for engine in (_capi.ENGINE_AUTO, _capi.ENGINE_TICK):
        lp, g = m.density_batch(np.asarray(qs), engine=engine)
        for c, (ref, ab) in enumerate(refs):
            got = np.concatenate([[lp[c]], g[c]])
            assert np.all(np.abs(got - ref) <= 1e-12 * ab + 1e-300), (engine, c)
    got = m.sample(cfg, seeds=seeds).chains                      # a 4-iteration static-HMC run, chain for chain
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
    m.close()


@pytest.mark.parametrize("split", [False, True])
def test_strict_location_scale_table_in_gather_mode_on_the_device(split):
    """alphas = Normal(mu, sd).latentVec(100): select sums folded, factors carried inside the selects (tests/test_emitter_host.py,
    test_strict_location_scale_table_in_gather_mode, is the host half)"""
    from rainier_amd import compute as CC
    from rainier_amd import modeling as M
    rng = np.random.default_rng(4)
    K, n = 100, 1500
    b = M.Normal(0, 1).latent
    alphas = M.Normal(M.Normal(0, 2).latent, M.Exponential(1).latent).latentVec(K)
    site = rng.integers(0, K, n).astype(float); x = rng.normal(size=n); ys = rng.poisson(3.0, n).astype(float)
    fn = lambda s, u: M.NegativeBinomial((CC.Lookup.apply(s, alphas) + b * u).logistic, 5.0)
    spec = M.Model.observe_vec(ys, [site, x], fn, split=split).compile("centred_table_100", inline=False)
    d = O.OracleDensity(spec)
    qs = [q for q in rng.normal(size=(6, spec.n_params)) * 0.3 if np.all(np.isfinite(d.update(q)))][:3]
    refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]
    m = R.Model(spec, device=0, **STRICT)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source
    lp, g = m.density_batch(np.asarray(qs), engine=_capi.ENGINE_TICK)
    for c, (ref, ab) in enumerate(refs):
        got = np.concatenate([[lp[c]], g[c]])
        assert np.all(np.abs(got - ref) <= 1e-12 * ab + 1e-300)
    m.close()


def test_strict_table_of_transformed_entries_on_the_device():
    """entries exp(z_k): the factor around an entry's select holds the entry's own parameter and is carried into the select reading it
    through the table (tests/test_emitter_host.py::test_strict_table_of_transformed_entries_in_gather_mode is the host half)"""
    from rainier_amd import compute as CC
    from rainier_amd import modeling as M
    rng = np.random.default_rng(5)
    K, n = 80, 900
    pre = M.Normal(0, 1).latent
    tab = [z.exp() for z in M.Normal(0, 0.3).latentVec(K)]
    site = rng.integers(0, K, n).astype(float); x = rng.normal(size=n); ys = rng.poisson(3.0, n).astype(float)
    fn = lambda s, u: M.NegativeBinomial((CC.Lookup.apply(s, tab) + pre * u).logistic, 5.0)
    spec = M.Model.observe_vec(ys, [site, x], fn, split=False).compile("exp_table", inline=False)
    d = O.OracleDensity(spec)
    qs = [q for q in rng.normal(size=(6, spec.n_params)) * 0.3 if np.all(np.isfinite(d.update(q)))][:3]
    refs = [d.update_both(np.asarray(q, dtype=np.float64)) for q in qs]
    m = R.Model(spec, device=0, **STRICT)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source
    lp, g = m.density_batch(np.asarray(qs), engine=_capi.ENGINE_TICK)
    for c, (ref, ab) in enumerate(refs):
        got = np.concatenate([[lp[c]], g[c]])
        assert np.all(np.abs(got - ref) <= 1e-12 * ab + 1e-300)
    m.close()
