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
