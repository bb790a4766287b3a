"""The reference's own benchmark suite (rainier-benchmark/.../bench/stan/*.scala) on the engine: each model in the reference's model
text (rainier_amd/models.py *_reference), on the reference's data (tests/golden/*.json), through rh_model_create's passes, then
`DensityFunction.update` for a batch of parameter vectors -- the operation the reference's JMH benchmarks time for ONE vector on
one JVM thread (BASELINE.md section 1 quotes their published p50).  Reported: microseconds per gradient at batch 1 (a launch and two
small copies: latency) and amortised over a large batch (throughput), wall clock around rh_density_eval_ex including its host
copies.  Usage: python tools/reference_benchmarks.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rainier_amd as R
from rainier_amd import _capi, models

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
load = lambda name: json.load(open(os.path.join(G, name)))
# (name, spec, the reference's published JMH figure in microseconds per gradient)
CASES = [
    ("eight_schools", models.eight_schools_reference(), 1.048),
    ("ar_k", models.ark_reference(load("ark.json")), 3.564),
    ("kidiq (N = 400)", models.kidiq_reference(load("kidiq.json")), 10.0),
    ("low_dim_gauss_mix", models.lowdim_gaussmix_reference(load("lowdim_gaussmix.json")), 649.555),
    ("glmm_poisson", models.glmm_poisson2_reference(100, 40, load("glmm_poisson2.json")), 235.011),
]
out = []
for name, spec, jmh_us in CASES:
    m = R.Model(spec, device=0, fp_contract=True, factor_outputs=True)
    rng = np.random.default_rng(1)
    row = {"model": name, "parameters": spec.n_params, "columns_in": len(spec.columns), "targets_in": len(spec.nrows),
           "reference_jmh_us_per_gradient_one_thread": jmh_us}
    for batch, engine, key in ((1, _capi.ENGINE_AUTO, "batch_1"), (16384, _capi.ENGINE_AUTO, "batch_16384"),
                               (16384, _capi.ENGINE_TICK, "batch_16384_tick_engine")):
        if engine == _capi.ENGINE_TICK and not spec.columns:
            continue
        q = rng.normal(size=(batch, spec.n_params)) * 0.3
        try:
            m.density_batch(q, engine=engine)                       # warm-up (module load, buffers)
            reps = 20 if batch == 1 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                m.density_batch(q, engine=engine)
            dt = (time.perf_counter() - t0) / reps
            row[key] = {"us_per_call": dt * 1e6, "us_per_gradient": dt * 1e6 / batch, "gradients_per_s": batch / dt}
        except Exception as e:  # noqa: BLE001
            row[key] = {"error": str(e)[:200]}
    print(json.dumps(row), flush=True)
    out.append(row)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
