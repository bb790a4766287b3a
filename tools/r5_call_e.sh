#!/bin/bash
# GPU call E of round 5: the whole GPU tier on the final tree, smoke(), the rocprofv3 evidence for the bench command (stats + counters
# -> profiles/r5_cfg2), the default bench line.  -> gpurun_out/r5_e/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_e; mkdir -p $O
ls rainier_amd/kcache | sort > $O/kcache_before.txt
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 1100 python -m pytest tests -m gpu -v --tb=short -rf -rs -p no:cacheprovider --durations=15 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|^SKIPPED|passed|failed" $O/tests.log | cut -c1-260 | tail -12
cp gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
( time timeout 120 python -c "import __graft_entry__ as G; G.smoke()" ) > $O/smoke.log 2>&1; grep -E "smoke ok|Error|rror" $O/smoke.log | head -3
bash tools/pmc_cfg2.sh ${1:-unknown} r5_e/cfg2 > $O/pmc_cfg2.log 2>&1; tail -12 $O/pmc_cfg2.log | cut -c1-300
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"))
    for k, v in d.get("configs", {}).items():
        r = v.get("roofline") or {}
        print("  ", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "mean_leapfrog_per_iteration", "seconds_total", "error", "skipped")}, r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
except Exception as e:
    print("bench output unreadable:", e)
PY
tail -3 $O/bench.err
mkdir -p $O/kcache_new; ls rainier_amd/kcache | sort > $O/kcache_after.txt
comm -13 $O/kcache_before.txt $O/kcache_after.txt | grep -v "\.tmp" | while read f; do cp -n rainier_amd/kcache/$f $O/kcache_new/ 2>/dev/null; done
ls $O/kcache_new | wc -l
