"""NUTS pinned DISTRIBUTIONALLY to a reference-pinned sampler.

The reference has no NUTS (its plugin point is sampler/Sampler.scala:52-62; BASELINE.json names NUTS for configurations 3-5), so
the sampler here is an extension whose bit-level parity is against the oracle's two statements of the algorithm --
both written in this repository.  What IS pinned to the reference is the EHMC path: `DefaultConfig` (sampler/Sampler.scala:17-27:
EHMCSampler(1024), DualAvgTuner(0.8), windowed diagonal mass) on the device is bit-identical to oracle/sampler.c, and that oracle
reproduces the reference's own SBC goldsets (tests/test_reference_goldset.py, TM/SBCModel.scala:46-267 at 1e-10).  This test ties the
two together from the outside: on models of the reference's own test / benchmark suites, in STRICT (JVM-faithful) builds of 1024
chains each,

    every parameter's posterior mean under NUTS(10) lies within 4 Monte-Carlo standard errors of its mean under DefaultConfig EHMC,
    the posterior variances agree within 10 %,
    and R-hat over the 2048 pooled chains (core/Trace.scala:52-120) is below 1.01 -- the two samplers' draws are one population.

A NUTS that selected leaves with the wrong weights, mis-stopped its doubling or broke detailed balance in its merges would move one
of these; EHMC with the same tuners is the yardstick because nothing about it is ours."""
import json
import os

import numpy as np
import pytest

import rainier_amd as R
from rainier_amd import _capi, models

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return json.load(open(os.path.join(G, name)))


def _moments(chains):
    """per parameter: mean, variance, Monte-Carlo standard error of the mean (sd / sqrt(ESS), ESS = Trace.diagnostics' formula)"""
    flat = chains.reshape(-1, chains.shape[-1])
    mean, var = flat.mean(axis=0), flat.var(axis=0)
    ess = np.array([e for _, e in R.diagnostics(chains)])
    return mean, var, np.sqrt(var / np.maximum(ess, 1.0))


CASES = {
    # name: (spec factory, warm-up, iterations, parameters held to the variance test, parameters the model reads through abs, build, chains)
    "eight_schools": (lambda: models.eight_schools_reference(), 400, 400, None, [], "strict", 1024),   # rainier-benchmark/.../bench/stan/EightSchools.scala
    # bench/stan/ARK.scala: an AR(5) series observed one value at a time (197 single-observation targets, lifted into one streamed target).
    # `sigma = Cauchy(0, 2.5).latent.abs` (ARK.scala:11): the density is symmetric in the raw parameter and a chain lives on one side
    # of zero, so the raw draws are two populations whatever the sampler (R-hat 19 under EHMC and NUTS alike, gpurun_out/r6_c); what
    # the model reads -- |parameter 1| -- is compared
    "ark": (lambda: models.ark_reference(_load("ark.json")), 400, 400, None, [1], "strict", 1024),
    # Neal's funnel (the README's plumbing model, cfg 1): x_i | v ~ N(0, e^{v/2}) has a log-normal scale mixture for a marginal --
    # its sample variance has no useful standard error at any affordable length -- so the variance test is held on v alone and
    # the x_i are compared through their means (0 by symmetry) and the pooled R-hat
    "funnel10": (lambda: models.funnel_reference(10), 600, 600, [0], [], "strict", 1024),
    # (bench/stan/GLMMPoisson2.scala, 146 parameters, was tried as a fourth case and is not one: at any length the GPU tier can afford
    #  NEITHER sampler converges on it -- 256 chains x (150 + 150) iterations, fast build: R-hat 45 for each sampler alone, NUTS at
    #  790 of its 1023 leapfrog steps per iteration, 6.5 minutes of device time -- so there are no two posteriors to compare
    #  (profiles/r6_side/glmm_timing.txt, profiles/r6_parity/nuts_distribution.txt))
}
BUILDS = {"strict": dict(math_mode=_capi.MATH_STRICT), "fast": dict(fp_contract=True, factor_outputs=True)}


@pytest.mark.parametrize("name", list(CASES))
def test_nuts_posterior_matches_default_config_ehmc(name):
    mk, warm, iters, var_params, folded, build, chains = CASES[name]
    spec = mk()
    m = R.Model(spec, device=0, **BUILDS[build])
    seeds_e = [910_000 + c for c in range(chains)]
    seeds_n = [920_000 + c for c in range(chains)]
    ehmc = m.sample(R.make_config(iters, warm), seeds=seeds_e)                        # DefaultConfig: EHMCSampler(1024) + DualAvg(0.8) + diag mass
    nuts = m.sample(R.make_config(iters, warm, R.NUTSSampler(10)), seeds=seeds_n)     # the same tuners, NUTS(10)
    ce, cn = ehmc.chains.copy(), nuts.chains.copy()
    for i in folded:
        ce[..., i] = np.abs(ce[..., i]); cn[..., i] = np.abs(cn[..., i])
    me, ve, se = _moments(ce)
    mn, vn, sn = _moments(cn)
    z = np.abs(mn - me) / np.sqrt(se ** 2 + sn ** 2 + 1e-300)
    ratio = vn / ve
    pooled = R.diagnostics(np.concatenate([ce, cn], axis=0))
    rhat = np.array([r for r, _ in pooled])
    own = max(max(r for r, _ in R.diagnostics(ce)), max(r for r, _ in R.diagnostics(cn)))
    worst = int(np.argmax(z))
    line = ("f2 %s: %d parameters, 2 x %d chains x %d draws: max |mean_nuts - mean_ehmc| = %.2f MCSE (parameter %d), variance ratio in [%.3f, %.3f], "
            "pooled R-hat max %.4f (each sampler alone: %.4f); leapfrog / iteration: ehmc %.1f, nuts %.1f" % (
                name, spec.n_params, chains, iters, z.max(), worst, ratio.min(), ratio.max(), rhat.max(), own,
                np.mean([s.leapfrogSteps for s in ehmc.stats]) / iters, np.mean([s.leapfrogSteps for s in nuts.stats]) / iters))
    print(line)
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "nuts_distribution.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert np.all(np.isfinite(nuts.chains)) and np.all(np.isfinite(ehmc.chains))
    assert z.max() < 4.0, line
    vsel = slice(None) if var_params is None else var_params
    assert np.all(ratio[vsel] > 0.9) and np.all(ratio[vsel] < 1.1), line
    assert rhat.max() < 1.01, line
    m.close()
