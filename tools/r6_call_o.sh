#!/bin/bash
# GPU call O of round 6: cfg 5 (centred) under NUTS with ONE early mass window and DefaultConfig's 50 iterations of step-size adaptation behind it
# (warm-up 100: window [20, 50)) -- does it converge at a third of the warm-up cost of DefaultConfig's 150?  -> gpurun_out/r6_o/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_o; mkdir -p $O
( time timeout 560 python bench.py --workload cfg5c --sampler default --steps 100 --warmup 100 --chains-per-gpu 1024 ) > $O/cfg5c_w100.json 2> $O/cfg5c_w100.err
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6_o")
try:
    d = json.loads([l for l in open(O + "/cfg5c_w100.json") if l.startswith("{")][-1]); r = d.get("roofline") or {}
    print("cfg5c NUTS warm-up 100 (window [20,50), skipLast 50): %.4g steps/s, warm-up %.1f s, timed %.1f s, %.3f ms/launch, frac %.3f (steady %s), rhat %.4f, ess/s %s, mean L %.1f, mass %s" % (
        d["value"], d["seconds_warmup"], d["seconds_timed"], r.get("avg_launch_ms", 0), r.get("frac", 0), (r.get("steady_state") or {}).get("frac"), d.get("rhat_max") or -1, d.get("ess_per_s"), d.get("mean_leapfrog_per_iteration", 0), d["config"]["mass"]))
except Exception as e:
    print("FAILED", e); print(open(O + "/cfg5c_w100.err").read()[-600:])
PY
