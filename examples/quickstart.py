"""The README quick start, runnable: `python examples/quickstart.py` on a machine with an MI355X."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import _capi, models
from rainier_amd.modeling import Model as RainierModel, Normal, Uniform

m = R.Model(models.linreg(n=1_000_000, k=3), fp_contract=True, factor_outputs=True)
cfg = R.make_config(iterations=200, warmupIterations=200, sampler=R.HMCSampler(32), stepSizeTuner=R.DualAvgTuner(0.8),
                    massMatrixTuner=R.IdentityMassMatrixTuner())
trace = m.sample(cfg, seeds=range(1024))
print("draws", trace.chains.shape, "rhat/ess", trace.diagnostics()[:2], trace.stats[0])
print("posterior mean", trace.chains.reshape(-1, 5).mean(axis=0), " (log sigma, a, b0, b1, b2; generating: log 0.7, 0.5, 1, -2, 0.5)")
print("L-BFGS optimum", m.optimize())

mu, sigma = Normal(0, 10).latent, Uniform(0, 1).latent
fit = RainierModel.observe([1.0, 2.0, 3.0], Normal(mu, sigma))
tr = R.Model(fit.compile()).sample(R.make_config(500, 500), seeds=[1, 2, 3, 4])
print("fit normal: mu", fit.predict(mu, tr.chains).mean(), "sigma", fit.predict(sigma, tr.chains).mean())

# the BASELINE models in the reference's own model text, lowered by the restated front end (compute.py): cfg 3 ...
es = R.Model(models.eight_schools_reference(), math_mode=_capi.MATH_STRICT)
tr = es.sample(R.make_config(500, 500), seeds=range(64))          # DefaultConfig: EHMC + DualAvg + windowed diagonal mass
print("eight schools (EightSchools.scala text): rhat max", max(r for r, _ in tr.diagnostics()))

# ... and Model.sample over several devices in ONE native call (here: two shards on device 0; the trace does not depend on it)
m2 = R.Model(models.linreg(n=1_000_000, k=3), fp_contract=True, factor_outputs=True)
t2 = R.sample_multi([m, m2], cfg, range(1024))
print("two shards == one:", np.array_equal(t2.chains, trace.chains))
