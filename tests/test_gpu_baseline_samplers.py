"""The SAMPLER configurations BASELINE.json names for cfg 3, cfg 4 and cfg 5, at the stated sizes (VERDICT r2 next #1a; the
density side of the same configurations is tests/test_gpu_baseline_sizes.py):

  cfg 3  eight schools, NUTS(max depth 10) + DefaultConfig's tuners (DualAvgTuner(0.8), DiagonalMassMatrixTuner(50, 1.5, 50, 50),
         sampler/Sampler.scala:17-27), 1024 chains per GPU -- the one-chain-per-wavefront variant of the sampler kernels (fewer
         than 4096 chains): 32 of the chains bit for bit against the oracle's nuts_iteration, every parameter R-hat < 1.01.
  cfg 4  logistic GLM, 50 covariates x 1e7 rows, NUTS(10) + windowed diagonal mass + DualAvg on the MFMA gradient kernel,
         256 chains: a short run at full size (the 2e5-row run of test_gpu_baseline_sizes.py checks convergence).
  cfg 5  hierarchical NegBin GLM, 10 000 groups x 100 observations (nVars 10 004, gather kernel, HBM-resident chain state),
         NUTS(10) with enough warm-up for the trees to leave max depth: finite draws, R-hat of the 4 shared parameters, mean
         tree depth reported.
The plugin point is sampler/Sampler.scala:52-62, the U-turn primitive sampler/LeapFrog.scala:35-47; NUTS itself is an
extension (the reference has none): parity is against the oracle's statement of the algorithm."""
import os
import time

import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi, models
from tests import oracle_lib as O
from tests.test_gpu_parity import _oracle_cfg

pytestmark = pytest.mark.gpu


def _note(line):
    print(line)
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "baseline_samplers.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def test_cfg3_eight_schools_nuts10_default_tuners_1024_chains():
    spec = models.eight_schools()
    chains, warm, iters = 1024, 300, 200
    cfg = R.make_config(iters, warm, R.NUTSSampler(10))      # DefaultConfig's DualAvgTuner(0.8) + DiagonalMassMatrixTuner(50, 1.5, 50, 50)
    seeds = [8000 + c for c in range(chains)]
    m = R.Model(spec, device=0, math_mode=_capi.MATH_STRICT)
    t0 = time.time()
    tr = m.sample(cfg, seeds=seeds)
    dt = time.time() - t0
    ocfg = _oracle_cfg(cfg, O.JM_DET)
    for c in list(range(16)) + list(range(chains - 16, chains)):          # both ends of the launch
        want, mass, st = O.sample_model(spec, ocfg, seeds[c])
        assert np.array_equal(tr.chains[c], want), (c, np.argwhere(tr.chains[c] != want)[:3])
        assert np.array_equal(tr.mass[c], mass) and tr.stats[c].leapfrogSteps == st.leapfrog_steps and tr.stats[c].stepSize == st.step_size
    diag = tr.diagnostics()
    rhat = max(r for r, _ in diag)
    steps = sum(st.leapfrogSteps for st in tr.stats)
    _note("cfg3 NUTS(10) 1024 chains x (%d + %d): rhat_max %.4f, ess_min %.0f, mean leapfrog / iteration %.1f, accept %.3f, %.2f s wall" % (
        warm, iters, rhat, min(e for _, e in diag), steps / (chains * iters), np.mean([st.meanAcceptProb for st in tr.stats]), dt))
    assert rhat < 1.01, rhat
    assert 0.7 < np.mean([st.meanAcceptProb for st in tr.stats]) < 0.92


def test_cfg4_sampler_config_at_1e7_rows_short():
    """measured (round 3): 40 + 8 iterations took 130 s, 24 + 4 took 67 s -- NUTS trees of depth 5-7 at 17 ms per leapfrog step -- so the run is cut
    to what shows the configuration WORKING at full size: the chains travel from their N(0,1) starts into the posterior's
    neighbourhood and the mass matrix is adapted.  Convergence proper (R-hat < 1.05, coefficients recovered) is asserted at
    2e5 rows by tests/test_gpu_baseline_sizes.py on the same kernels."""
    n, k, chains = 10_000_000, 50, 256
    spec = models.logistic(n=n, k=k)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "rh_grad_glm_kernel" in m.hip_source
    warm, iters = 14, 3
    cfg = R.make_config(iters, warm, R.NUTSSampler(10), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(6, 1.5, 4, 2), engine=_capi.ENGINE_TICK)
    s = R.Sampler(m, cfg, [4000 + c for c in range(chains)])
    t0 = time.time(); s.warmup(); s.run(iters); dt = time.time() - t0
    assert s.timing()["dominant_kernel"] == "rh_grad_glm_kernel"
    draws = s.draws()
    stats, mass = s.stats()
    s.close(); m.close()
    assert np.all(np.isfinite(draws)) and not np.allclose(mass, 1.0)
    rng = np.random.default_rng(4); rng.standard_normal((k, n)); beta_true = rng.standard_normal(k)
    post_mean = draws.reshape(-1, k + 1).mean(axis=0)
    dev = float(np.max(np.abs(post_mean[1:] - beta_true)))
    lf, wlf = sum(st.leapfrogSteps for st in stats), sum(st.warmupLeapfrogSteps for st in stats)
    _note("cfg4 NUTS(10) + diag mass at 1e7 x 50, 256 chains x (%d + %d): mean leapfrog / iteration %.1f (warm-up %.1f), accept %.3f, "
          "max |posterior mean - beta| %.2e, %.1f s wall" % (warm, iters, lf / (chains * iters), wlf / (chains * warm),
                                                              np.mean([st.meanAcceptProb for st in stats]), dev, dt))
    # the starts are N(0,1) draws (|start - beta| ~ 1.4 on average, ~ 2000 posterior standard deviations)
    assert abs(post_mean[0]) < 0.3 and dev < 0.3, dev      # (24 + 4 iterations reached 2.9e-2, 40 + 8: 1.4e-2)


def test_cfg5_hier_negbin_nuts10_at_full_size():
    """What was measured in round 3 with 150 + 40 iterations (64 chains, 265 s): the trees do NOT leave max depth (1023 leaves in
    the sampling phase, 745 on average during warm-up, acceptance 0.94, R-hat 3-6 on m and s).  cfg 5 as BASELINE.json writes it is
    the NON-centred hierarchy (alpha_g = 10 m + e^s z_g) under 100 observations per group: the data pin every alpha_g, so z_g, m
    and s are tied along narrow ridges that a diagonal mass matrix cannot straighten -- a property of the posterior, not of the
    engine (the centred form of the same model is what the data call for).  So this test runs the configuration at full size for
    a bounded time and asserts what the engine owes: finite draws, the gather kernel + HBM-resident state under NUTS, an adapted
    mass matrix, leapfrog accounting; tree depth, R-hat and cost are REPORTED (gpurun_out/baseline_samplers.txt)."""
    G, per, chains = 10_000, 100, 64
    spec = models.hier_negbin(G, per)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "rh_grad_gather_kernel" in m.hip_source and "#define RH_BIGN 1" in m.hip_source
    warm, iters = 36, 6
    cfg = R.make_config(iters, warm, R.NUTSSampler(10), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(10, 1.5, 10, 6))
    s = R.Sampler(m, cfg, [9000 + c for c in range(chains)])
    t0 = time.time(); s.warmup(); tw = time.time() - t0
    t0 = time.time(); s.run(iters); dt = time.time() - t0
    tim = s.timing()
    draws = s.draws()
    stats, mass = s.stats()
    s.close()
    assert tim["dominant_kernel"] == "rh_grad_gather_kernel"
    assert draws.shape == (chains, iters, spec.n_params) and np.all(np.isfinite(draws))
    assert all(1 <= st.leapfrogSteps <= iters * 1023 and st.warmupLeapfrogSteps >= warm for st in stats)
    assert not np.allclose(mass, 1.0) and np.all(mass > 0)
    rhat = [r for r, _ in R.diagnostics(draws[:, :, :4])]
    lf = sum(st.leapfrogSteps for st in stats) / (chains * iters)
    wlf = sum(st.warmupLeapfrogSteps for st in stats) / (chains * warm)
    _note("cfg5 NUTS(10) 10 000 x 100, %d chains x (%d + %d): mean leapfrog / iteration %.1f (tree depth ~%.1f; warm-up %.1f), accept %.3f, "
          "rhat of the 4 shared parameters %s, warm-up %.1f s, run %.1f s (%.2f ms per leapfrog step)" % (
              chains, warm, iters, lf, np.log2(lf + 1), wlf, np.mean([st.meanAcceptProb for st in stats]),
              ["%.3f" % r for r in rhat], tw, dt, dt / (lf * iters) * 1e3))


def test_cfg5_centred_parameterisation_in_gather_mode():
    """The same model with the group effects themselves as parameters, alpha_g ~ Normal(mu, e^s) (models.hier_negbin_centred: what
    Real.parameter { a => Normal(mu, sigma).logDensity(a) } gives, compute/Real.scala:63-78): the prior ties every table entry to
    the shared parameters, the reference hands d/d mu and d/d s of it over as sums over all 10 000 entries.  Fast builds lift the
    prior into a row target over the group index (csrc/lift.cpp) and run in gather mode like the non-centred form.
      * 10 000 groups x 10 observations: (logp, gradient) against the oracle evaluating the ORIGINAL program;
      * 10 000 x 100, NUTS(10) + windowed diagonal mass, 64 chains: this is the form whose posterior NUTS can traverse -- tree
        depth and R-hat are reported next to the non-centred run above."""
    spec = models.hier_negbin_centred(10_000, 10)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    assert "#define RH_HAS_GATHER 1\n" in m.hip_source and "#define RH_BIGN 1" in m.hip_source and "#define RH_NSHARED 4\n" in m.hip_source
    rng = np.random.default_rng(77)
    q = rng.normal(size=(3, spec.n_params)) * 0.3
    lp, g = m.density_batch(q)
    d = O.OracleDensity(spec)
    for c in range(2):
        ref, ab = d.update_both(q[c])
        got = np.concatenate([[lp[c]], g[c]])
        err = np.abs(got - ref) / (1e-11 * ab + 1e-300)
        assert np.all(err <= 1.0), (c, float(err.max()), int(np.argmax(err)))
    m.close()
    G, per, chains = 10_000, 100, 64
    spec = models.hier_negbin_centred(G, per)
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    warm, iters = 60, 12         # (80 + 20 measured: R-hat 0.995 .. 1.007, means 0.299 / -0.200 / 1.011 / -0.689, 50 s)
    cfg = R.make_config(iters, warm, R.NUTSSampler(10), R.DualAvgTuner(0.8), R.DiagonalMassMatrixTuner(15, 1.5, 15, 10))
    s = R.Sampler(m, cfg, [9500 + c for c in range(chains)])
    t0 = time.time(); s.warmup(); tw = time.time() - t0
    t0 = time.time(); s.run(iters); dt = time.time() - t0
    draws = s.draws()
    stats, mass = s.stats()
    s.close()
    assert np.all(np.isfinite(draws)) and not np.allclose(mass, 1.0)
    rhat = [r for r, _ in R.diagnostics(draws[:, :, :4])]
    lf = sum(st.leapfrogSteps for st in stats) / (chains * iters)
    wlf = sum(st.warmupLeapfrogSteps for st in stats) / (chains * warm)
    post = draws[:, :, :4].reshape(-1, 4).mean(axis=0)
    _note("cfg5 CENTRED NUTS(10) 10 000 x 100, %d chains x (%d + %d): mean leapfrog / iteration %.1f (tree depth ~%.1f; warm-up %.1f), accept %.3f, "
          "rhat of (b0, b1, mu, s) %s, posterior means %s, warm-up %.1f s, run %.1f s (%.2f ms per leapfrog step)" % (
              chains, warm, iters, lf, np.log2(lf + 1), wlf, np.mean([st.meanAcceptProb for st in stats]),
              ["%.3f" % r for r in rhat], np.array2string(post, precision=3), tw, dt, dt / (lf * iters) * 1e3))
    # the data were generated with b = (0.3, -0.2), alpha_g ~ N(1, 0.5): mu -> 1, s -> log 0.5
    assert abs(post[0] - 0.3) < 0.05 and abs(post[1] + 0.2) < 0.05 and abs(post[2] - 1.0) < 0.1 and abs(post[3] - np.log(0.5)) < 0.1, post
    # ... and the chains agree with each other: R-hat of the four shared parameters over 64 chains (measured 0.995 .. 1.017)
    assert max(rhat) < 1.05, rhat
