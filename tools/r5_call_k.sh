#!/bin/bash
# GPU call K of round 5: the final check -- the whole GPU tier and the default bench line on the tree as it is handed over.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_k; mkdir -p $O
ls rainier_amd/kcache | sort > $O/kcache_before.txt
( time timeout -s INT --kill-after=60 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/tests.log 2>&1
tail -5 $O/tests.log | cut -c1-300
( time timeout 120 python -c "import __graft_entry__ as G; G.smoke()" ) > $O/smoke.log 2>&1; grep -E "smoke ok|rror" $O/smoke.log | head -2
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"))
    print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:140])
    for k, v in d.get("configs", {}).items():
        r = v.get("roofline") or {}
        print("  ", k, {kk: v.get(kk) for kk in ("value", "mean_leapfrog_per_iteration", "seconds_total", "error")}, r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
except Exception as e:
    print("bench output unreadable:", e)
PY
tail -3 $O/bench.err
mkdir -p $O/kcache_new; ls rainier_amd/kcache | sort > $O/kcache_after.txt
comm -13 $O/kcache_before.txt $O/kcache_after.txt | grep -v "\.tmp" | while read f; do cp -n rainier_amd/kcache/$f $O/kcache_new/ 2>/dev/null; done
ls $O/kcache_new | wc -l
