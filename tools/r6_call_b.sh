#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# GPU call B of round 6: the whole GPU tier at HEAD (live-chain lists, gradient-only requests, the fallback gather walks without a
# divergent ragged tile), then the bench's side legs as the `configs` block runs them.  -> gpurun_out/r6_b/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_b; mkdir -p $O
( time timeout -s INT 2400 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider --durations=15 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests.log | tail -25
for leg in "cfg5 hmc8 4 2 1024" "cfg4 hmc8 2 2 256" "cfg2d default 64 200 1024" "cfg5c default 60 36 1024" "cfg4 default 100 60 256"; do
  set -- $leg
  ( timeout 900 python bench.py --workload $1 --sampler $2 --steps $3 --warmup $4 --chains-per-gpu $5 ) > $O/leg_$1_$2.json 2> $O/leg_$1_$2.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/leg_$1_$2.json") if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("$1 $2: %.4g steps/s, %.2f s timed, %.2f s warm-up, %s %.3f ms/launch x %d, frac %.3f, slot eff %.3f, steady %s, rhat %s, ess/s %s, mean L %.1f" % (
        d["value"], d["seconds_timed"], d["seconds_warmup"], r.get("kernel"), r.get("avg_launch_ms", 0), r.get("launches", 0), r.get("frac", 0),
        r.get("slot_efficiency", 0), (r.get("steady_state") or {}).get("frac"), d.get("rhat_max"), d.get("ess_per_s"), d.get("mean_leapfrog_per_iteration", 0)))
except Exception as e:
    print("$1 $2: FAILED", e); print(open("$O/leg_$1_$2.err").read()[-1500:])
PY
done
for v in 1 0; do
  ( RH_VALUE_FREE=$v timeout 600 python bench.py --workload cfg5 --sampler hmc8 --steps 4 --warmup 2 --chains-per-gpu 1024 ) > $O/vf$v.json 2>/dev/null
  python -c "
import json; d = json.loads([l for l in open('$O/vf$v.json') if l.startswith('{')][-1]); r = d['roofline']
print('cfg5 hmc8 RH_VALUE_FREE=$v: %.3f ms/launch, frac %.3f' % (r['avg_launch_ms'], r['frac']))"
  ( RH_VALUE_FREE=$v timeout 600 python bench.py --workload cfg4 --sampler hmc8 --steps 2 --warmup 2 --chains-per-gpu 256 ) > $O/vf4_$v.json 2>/dev/null
  python -c "
import json; d = json.loads([l for l in open('$O/vf4_$v.json') if l.startswith('{')][-1]); r = d['roofline']
print('cfg4 hmc8 RH_VALUE_FREE=$v: %.3f ms/launch, frac %.3f' % (r['avg_launch_ms'], r['frac']))"
done
