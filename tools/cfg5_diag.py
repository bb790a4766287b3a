"""cfg 5 at full size: do chains that were given the SAME parameter vector get the same (logp, gradient) bits from the gather path?
(tests/test_gpu_baseline_sizes.py asserts it; this prints WHERE they differ when they do: which chains -- position in the K-group --,
which outputs, how far.)  usage: [RH_* env] python tools/cfg5_diag.py [chains] [splits]"""
import os, sys
os.environ.setdefault("RH_DIAG", "1")   # experiment switches are read only in a process that asks for them (csrc/rir.hpp: rh::knob)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import models, _capi
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
spec = models.hier_negbin(10_000, 100)
m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
distinct = np.random.default_rng(55).normal(size=(4, spec.n_params)) * 0.3
idx = np.random.default_rng(5).permutation(np.arange(chains) % 4)
q = np.ascontiguousarray(distinct[idx])
lp, g = m.density_batch(q, grad_splits=splits)
got = np.concatenate([lp[:, None], g], axis=1)
env = {k: v for k, v in os.environ.items() if k.startswith("RH_")}
for j in range(4):
    rows = got[idx == j]; who = np.flatnonzero(idx == j)
    ref = rows[0]
    bad = np.flatnonzero(np.any(rows != ref, axis=1))
    if len(bad) == 0:
        print(env, "vector", j, ": all", len(rows), "chains identical"); continue
    cols = np.flatnonzero(np.any(rows != ref, axis=0))
    rel = np.max(np.abs(rows[bad][:, cols] - ref[cols]) / (np.abs(ref[cols]) + 1e-300))
    print(env, "vector", j, ":", len(bad), "of", len(rows), "chains differ from the first; chain ids", who[bad][:12], "kk", (who[bad] % 4)[:12],
          "| outputs", cols[:8], "... (", len(cols), "of", got.shape[1], ") | max rel diff %.3g" % rel)
m.close()
