"""cfg 2 with the 1024 chains cut into S shards that run CONCURRENTLY on one device (S model handles = S streams, one host
thread each, rh_sample_multi): while one shard's latency-bound rh_tick_kernel runs, the other shards' gradient kernels keep
the GPU busy.  Prints ms per HMC iteration for S = 1, 2, 4."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import models, _capi
spec = models.linreg(n=1_000_000, k=3)
iters, warm, L, chains = 40, 8, 32, 1024
cfg = R.make_config(iters, warm, R.HMCSampler(L), R.DualAvgTuner(0.8), R.IdentityMassMatrixTuner())
seeds = [1000 + c for c in range(chains)]
ms = [R.Model(spec, device=0, fp_contract=True, factor_outputs=True) for _ in range(4)]
base = None
for S in (1, 2, 4, 1, 2):
    t = time.perf_counter()
    tr = R.sample_multi(ms[:S], cfg, seeds)
    dt = time.perf_counter() - t
    if base is None: base = tr.chains
    print(json.dumps({"shards": S, "total_s": dt, "ms_per_iteration_incl_warmup": dt / (iters + warm) * 1e3,
                      "identical_to_unsharded": bool(np.array_equal(tr.chains, base))}), flush=True)
