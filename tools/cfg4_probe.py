import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
import numpy as np
import rainier_amd as R
from rainier_amd import models, _capi
n = int(sys.argv[1]); chains = int(sys.argv[2]); eng = int(sys.argv[3])
reference = len(sys.argv) > 4 and sys.argv[4] == "reference"
t = time.time()
if reference:
    # cfg 4 exactly as the JVM would hand it over: the reference's model text through Model.observe (8-way split), its own
    # algebra and gradient -> 1945 columns in two row targets; rh_model_create folds it back
    from rainier_amd import modeling as M
    cols = models.logistic_data(n, 50)
    a = M.Normal(0, 1).latent; bs = M.Normal(0, 1).latentVec(50)
    spec = M.Model.observe_vec(cols[0], cols[1:], lambda *u: M.Bernoulli((a + M.Real.sum([ui * bi for ui, bi in zip(u, bs)])).logistic),
                               split=True).compile("cfg4_reference_%d" % n)
    print('reference lowering: %d columns, rows per target %s' % (len(spec.columns), spec.nrows), flush=True)
else:
    spec = models.logistic(n=n, k=50)
print('data', time.time() - t, flush=True)
t = time.time(); m = R.Model(spec, device=0, factor_outputs=True, fp_contract=True); print('model', time.time() - t, flush=True)
cfg = R.make_config(2, 0, R.HMCSampler(4), R.StaticStepSize(1e-4), R.IdentityMassMatrixTuner(), engine=eng)
s = R.Sampler(m, cfg, list(range(chains)))
s.warmup(); s.timing(reset=True)
t = time.time(); s.run(2); dt = time.time() - t
tim = s.timing()
print(json.dumps({"n": n, "chains": chains, "engine": eng, "form": "reference" if reference else "natural", "s_per_tick": dt / 8, "row_chain_evals_per_s": n * chains * 8 / dt, "kernel_ms": tim["kernel_ms"], "launches": tim["launches"], "kernel": tim["dominant_kernel"]}))
