#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# GPU call E of round 6: after the explicit-fusion row code and the GLM kernel's hoisted gradient-only choice -- the tests that failed in
# call D first, the cfg 4 A/B, then the whole -m gpu tier (its new code objects harvested for the in-tree cache), two side legs.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_e; mkdir -p $O
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 600 python -m pytest tests/test_gpu_live_chains.py tests/test_gpu_nuts_distribution.py -m gpu -q --tb=short -rf -p no:cacheprovider --durations=8 --timeout 420 ) > $O/tests_first.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests_first.log | tail -12; grep "^f2 " $O/tests_first.log | cut -c1-400
bash tools/r6_cfg4_ab.sh r6_e
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider --durations=30 --timeout 420 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests.log | tail -20
unset RH_DIAG
for leg in "cfg5 hmc8 4 2 1024" "cfg2d default 256 200 1024"; do
  set -- $leg
  ( timeout 300 python bench.py --workload $1 --sampler $2 --steps $3 --warmup $4 --chains-per-gpu $5 ) > $O/leg_$1_$2.json 2> $O/leg_$1_$2.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/leg_$1_$2.json") if l.startswith("{")][-1]); r = d.get("roofline") or {}
    print("$1 $2: %.4g steps/s, %.2f s timed, %s %.3f ms/launch x %d, frac %.3f, steady %s, rhat %s, ess/s %s" % (d["value"], d["seconds_timed"], r.get("kernel"), r.get("avg_launch_ms", 0), r.get("launches", 0), r.get("frac", 0), (r.get("steady_state") or {}).get("frac"), d.get("rhat_max"), d.get("ess_per_s")))
except Exception as e:
    print("$1 $2: FAILED", e); print(open("$O/leg_$1_$2.err").read()[-800:])
PY
done
