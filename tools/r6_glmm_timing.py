"""What does a DefaultConfig EHMC / NUTS(10) run of bench/stan/GLMMPoisson2 cost on the device, per build flavour and chain count?  (sizing of
tests/test_gpu_nuts_distribution.py::glmm_poisson2: the strict build at 1024 chains did not finish in 25 minutes, gpurun_out/r6_c)
usage (GPU box): python tools/r6_glmm_timing.py"""
import json, os, sys, time
os.environ.setdefault("RH_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import _capi, models

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
spec = models.glmm_poisson2_reference(100, 40, json.load(open(os.path.join(G, "glmm_poisson2.json"))))
for label, build in (("fast", dict(fp_contract=True, factor_outputs=True)), ("strict", dict(math_mode=_capi.MATH_STRICT))):
    t0 = time.perf_counter()
    m = R.Model(spec, device=0, **build)
    print(label, "create %.1f s" % (time.perf_counter() - t0), m.engines(), flush=True)
    for chains in (64, 256):
        for sname, smp in (("ehmc", None), ("nuts10", R.NUTSSampler(10))):
            cfg = R.make_config(30, 30, smp) if smp is not None else R.make_config(30, 30)
            t0 = time.perf_counter()
            tr = m.sample(cfg, seeds=[100 + c for c in range(chains)])
            dt = time.perf_counter() - t0
            lf = np.mean([s.leapfrogSteps + s.warmupLeapfrogSteps for s in tr.stats])
            print("  %s %s %d chains, 30 + 30 iterations: %.1f s, %.0f leapfrog steps per chain (%.2f ms per step of the slowest chain)" % (
                label, sname, chains, dt, lf, 1e3 * dt / max(1, max(s.leapfrogSteps + s.warmupLeapfrogSteps for s in tr.stats))), flush=True)
            if dt > 120:
                break
    m.close()
