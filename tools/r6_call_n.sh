#!/bin/bash
# GPU call N of round 6: `bench.py --long-configs` on the tree as handed over (cfg 5's converging NUTS leg: warm-up 150 + 400 iterations) -> gpurun_out/r6_n/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_n; mkdir -p $O
( time timeout 1750 python bench.py --gpus 1 --steps 20 --warmup 5 --long-configs ) > $O/bench_long.json 2> $O/bench_long.err; tail -4 $O/bench_long.err
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6_n")
d = json.loads([l for l in open(os.path.join(O, "bench_long.json")) if l.startswith("{")][-1])
r = d["roofline"]; print("cfg2: %.4g steps/s, %.3f ms/step, %s %.4f ms/launch, frac %.4f" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"]))
for k, v in (d.get("configs") or {}).items():
    rr = v.get("roofline") or {}
    print(" ", k, "->", ("%.4g steps/s, %s %.3f ms/launch, frac %.3f, steady %s, rhat %s, ess/s %s, mean L %.1f, %.0f s" % (v["value"], rr.get("kernel"), rr.get("avg_launch_ms", 0), rr.get("frac", 0), (rr.get("steady_state") or {}).get("frac"), v.get("rhat_max"), v.get("ess_per_s"), v.get("mean_leapfrog_per_iteration") or 0, v.get("seconds_total", 0))) if "value" in v else v)
PY
