"""The reference's test models written the way the reference writes them (rainier_amd/modeling.py), lowered
through the restated rainier-compute front half (rainier_amd/compute.py: Real normal forms, Gradient, TargetGroup.inlinable +
PartialEvaluator.inline, Translator) to the RIR Compiler.compileTargets would produce, and run end to end: the eleven SBC
goldsets of core/SBCTest.scala:19-34 and OptimizerTest's "fit normal"."""
import json
import os

import numpy as np
import pytest

from rainier_amd import models
from rainier_amd.modeling import (SBC, Bernoulli, Binomial, Exponential, Gamma, Geometric, Laplace, LogNormal, Model,
                                  NegativeBinomial, Normal, Poisson, Uniform, JavaRandom)
from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ALL = json.load(open(os.path.join(HERE, "golden", "sbc_goldsets.json")))

# rainier-test/src/main/scala/com/stripe/rainier/core/SBCModel.scala:46-267, definition for definition
SBC_MODELS = {
    "SBCUniformNormal": lambda: SBC([Uniform(0, 1)], lambda x: Normal(x, 1)),
    "SBCLogNormal": lambda: SBC([LogNormal(0, 1)], lambda x: LogNormal(x, x)),
    "SBCExponential": lambda: SBC([LogNormal(0, 1)], lambda x: Exponential(x)),
    "SBCLaplace": lambda: SBC([LogNormal(0, 1)], lambda x: Laplace(x, x)),
    "SBCGamma": lambda: SBC([LogNormal(0, 1)], lambda x: Gamma(x, x)),
    "SBCBernoulli": lambda: SBC([Uniform(0, 1)], lambda x: Bernoulli(x)),
    "SBCBinomial": lambda: SBC([Uniform(0, 1)], lambda x: Binomial(x, 10)),
    "SBCGeometric": lambda: SBC([Uniform(0, 1)], lambda x: Geometric(x)),
    "SBCNegativeBinomial": lambda: SBC([Uniform(0, 1)], lambda x: NegativeBinomial(x, 10)),
    "SBCBinomialPoissonApproximation": lambda: SBC([Uniform(0, 0.04)], lambda x: Binomial(x, 200)),
    "SBCLargePoisson": lambda: SBC([Uniform(0.8, 1)], lambda x: Poisson(x * 1000)),
}


@pytest.mark.parametrize("name", sorted(SBC_MODELS))
def test_sbc_goldset_through_the_modelling_api(oracle, name):
    # SBCModel.scala:33-39: synthesize and sample from ONE ScalaRNG(1528673302081L) stream, HMCSampler(1), DualAvgTuner(0.8)
    sbc = SBC_MODELS[name]()
    rng = O.JavaRandom(ALL["seed"])
    values, truth = sbc.synthesize(ALL["synthetic_samples"], rng)
    model, real = sbc.fit(values)
    spec = model.compile(name)
    # Model.observe cuts the 1000 points into 8 + 8 x 124 (core/Model.scala:98-132): two likelihood targets.  Nine of the
    # eleven likelihoods are "inlinable" (Target.scala:136-207) and are folded into O(1) data-free targets at compile time,
    # exactly what the reference runs; Gamma and Laplace stream their rows (the second target reads 8 columns per row).
    assert spec.n_params == 1 and len(spec.nrows) == 3
    assert spec.nrows == ([0, 8, 124] if name in ("SBCGamma", "SBCLaplace") else [0, 0, 0]), spec.nrows
    gold = np.array(ALL["models"][name]["goldset"])
    cfg = O.make_config(sampler=O.HMC, n_steps=1, iterations=len(gold), warmup=ALL["warmup"], step_tuner=O.STEP_DUALAVG,
                        delta=0.8, mass_tuner=O.MASS_IDENTITY, math_mode=O.JM_LIBM)
    d = O.OracleDensity(spec, O.JM_LIBM)
    draws, _, st, rc = O.sample_chain_state(d.fn_ptr, d.handle, 1, cfg, rng.r)
    got = model.predict(real, draws)
    assert rc == 0 and np.abs((got - gold) / gold).max() < 1e-12, (name, got[:3], gold[:3])
    if name == "SBCGamma":            # no exp/log of a parameter-dependent quantity that glibc and HotSpot round differently:
        assert np.array_equal(got, gold)   # the reference's recorded draws, reproduced bit for bit through the restated pipeline
    # the same model with inlining switched off streams every row and lands on the same draws
    spec2 = model.compile(name + "_streamed", inline=False)
    assert spec2.nrows == [0, 8, 124]
    rng2 = O.JavaRandom(ALL["seed"]); sbc.synthesize(ALL["synthetic_samples"], rng2)
    d2 = O.OracleDensity(spec2, O.JM_LIBM)
    draws2, _, _, rc2 = O.sample_chain_state(d2.fn_ptr, d2.handle, 1, cfg, rng2.r)
    assert rc2 == 0 and np.abs((model.predict(real, draws2) - gold) / gold).max() < 1e-10, name


def test_fit_normal_through_the_modelling_api(oracle):
    # optimizer/OptimizerTest.scala:8-13
    mu = Normal(0, 10).latent
    sigma = Uniform(0, 1).latent
    m = Model.observe([1.0, 2.0, 3.0], Normal(mu, sigma))
    spec = m.compile("fit_normal")
    assert spec.n_params == 2
    ref = models.fit_normal()
    q = np.array([0.2, 0.7])
    a, b = O.OracleDensity(spec).update(q), O.OracleDensity(ref).update(q)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    x, evals = O.optimize_model(spec)
    assert 0 < evals < 40 and abs(float(m.predict(mu, x)) - 2.0) < 0.02


def test_pure_python_java_random_matches_published_values():
    assert JavaRandom(0).next_double() == 0.730967787376657
    assert abs(JavaRandom(0).next_gaussian() - 0.8025330637390305) < 1e-15
