"""Per-iteration vs per-leapfrog-step cost of the chain engine on a data-free model: t(L) = a + b L."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import models, _capi
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
name = sys.argv[3] if len(sys.argv) > 3 else "funnel"
spec = getattr(models, name)()
m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
for L in (1, 5, 50, 500):
    cfg = R.make_config(iters, 0, R.HMCSampler(L), R.StaticStepSize(0.05), R.IdentityMassMatrixTuner())
    s = R.Sampler(m, cfg, list(range(chains)))
    s.warmup()
    t = time.perf_counter(); s.run(iters); dt = time.perf_counter() - t
    print(json.dumps({"model": name, "chains": chains, "L": L, "us_per_iteration": dt / iters * 1e6, "us_per_leapfrog": dt / iters / L * 1e6,
                      "steps_per_s": chains * iters * L / dt}), flush=True)
    s.close()
