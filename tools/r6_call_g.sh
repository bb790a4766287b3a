#!/bin/bash
# GPU call G of round 6: does BASELINE cfg 4 (NUTS + diagonal mass, 1e7 x 50, 256 chains) converge sooner with a warm-up long enough for the
# mass windows to see more than 15 draws?  Two warm-up lengths, 60 timed iterations each.  -> gpurun_out/r6_g/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_g; mkdir -p $O
for W in 100 150; do
  ( time timeout 600 python bench.py --workload cfg4 --sampler default --steps 60 --warmup $W --chains-per-gpu 256 ) > $O/cfg4_nuts_w$W.json 2> $O/cfg4_nuts_w$W.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/cfg4_nuts_w$W.json") if l.startswith("{")][-1]); r = d.get("roofline") or {}
    print("cfg4 NUTS warm-up $W: %.4g steps/s, warm-up %.1f s, timed %.1f s, %.3f ms/launch, frac %.3f (steady %s), rhat %.4f, ess/s %s, mean L %.1f, mass %s" % (
        d["value"], d["seconds_warmup"], d["seconds_timed"], r.get("avg_launch_ms", 0), r.get("frac", 0), (r.get("steady_state") or {}).get("frac"), d.get("rhat_max") or -1, d.get("ess_per_s"), d.get("mean_leapfrog_per_iteration", 0), d["config"]["mass"]))
except Exception as e:
    print("cfg4 NUTS warm-up $W: FAILED", e); print(open("$O/cfg4_nuts_w$W.err").read()[-600:])
PY
done
