#!/bin/bash
# GPU call K of round 6: side figures for DESIGN 3.8 -- the judged configuration in a JVM-faithful (strict) build, and the data-free configurations
# at 32 768 chains (where they stop being one wavefront per SIMD).  -> gpurun_out/r6_k/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_k; mkdir -p $O
( timeout 400 python bench.py --strict --steps 20 --warmup 5 --no-cpu-baseline --no-ess --no-inlined --no-configs --no-live-traffic ) > $O/bench_cfg2_strict.json 2> $O/bench_cfg2_strict.err
python bench.py --workload cfg1 --no-cpu-baseline --steps 400 --warmup 200 --chains-per-gpu 32768 > $O/cfg1_32768.json 2>&1
python bench.py --workload cfg3 --no-cpu-baseline --steps 400 --warmup 200 --chains-per-gpu 32768 > $O/cfg3_ehmc_32768.json 2>&1
python bench.py --workload cfg3 --sampler nuts --no-cpu-baseline --steps 200 --warmup 200 --chains-per-gpu 32768 > $O/cfg3_nuts_32768.json 2>&1
python - <<'PY'
import json, os, glob
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6_k")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d.get("roofline") or {}
        print(os.path.basename(f), "%.4g steps/s, %.3f ms/step, %s %.4f ms/launch, frac %.3f, rhat %s, ess/s %s" % (d["value"], d["ms_per_step"], r.get("kernel"), r.get("avg_launch_ms") or 0, r.get("frac") or 0, d.get("rhat_max"), d.get("ess_per_s")))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
