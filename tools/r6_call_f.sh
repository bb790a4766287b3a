#!/bin/bash
# GPU call F of round 6: the tree as handed over -- the whole -m gpu tier against the driver's 1200 s limit (cache harvested in call E), smoke(),
# the rocprofv3 evidence of the bench command (stats + PMC passes, sha-stamped), the driver's bench command itself.  -> gpurun_out/r6_f/, r6_cfg2/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_f; mkdir -p $O
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 1300 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider --durations=25 --timeout 420 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests.log | tail -20; ls $O/kcache_new 2>/dev/null | wc -l
( time timeout 120 python -c "import __graft_entry__ as G; G.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
( timeout 900 bash tools/pmc_cfg2.sh ${1:-unknown} r6_cfg2 ) 2>&1 | tail -12
cd "$GRAFT_REPO_ROOT"
( time timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r = d["roofline"]; print("cfg2: %.4g steps/s, %.3f ms/step, %s %.4f ms/launch, frac %.4f, traffic %s" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], r.get("traffic")))
print("cpu_baseline:", json.dumps(d.get("cpu_baseline"))[:400])
for k, v in (d.get("configs") or {}).items():
    rr = v.get("roofline") or {}
    print(" ", k, "->", ("%.4g steps/s, %s %.3f ms/launch, frac %.3f, steady %s, rhat %s, ess/s %s, %.0f s" % (v["value"], rr.get("kernel"), rr.get("avg_launch_ms", 0), rr.get("frac", 0), (rr.get("steady_state") or {}).get("frac"), v.get("rhat_max"), v.get("ess_per_s"), v.get("seconds_total", 0))) if "value" in v else v)
PY
