#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# GPU call A of round 6: the live-chain compaction (rh_compact_kernel, list-addressed gradient launches and ticks): its own tests, the
# tests of everything it touched (fused launches, multi-shard, tick-engine parity, gather mode, GLM), then the side legs of the bench
# that it is for (cfg 2 under DefaultConfig, cfg 4 / cfg 5 under NUTS) with and without compaction.  -> gpurun_out/r6_a/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_a; mkdir -p $O
( time timeout -s INT 1500 python -m pytest tests/test_gpu_live_chains.py tests/test_gpu_fused.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py \
    -m gpu -q --tb=short -rf -p no:cacheprovider --durations=12 -x \
    -k "not torch_distributed" ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests.log | tail -15
for leg in "cfg2d default 32 100 1024" "cfg4 default 24 24 256" "cfg5c default 12 18 1024" "cfg5 hmc8 4 2 1024" "cfg4 hmc8 2 2 256"; do
  set -- $leg
  for c in 1 0; do
    ( RH_COMPACT=$c timeout 600 python bench.py --workload $1 --sampler $2 --steps $3 --warmup $4 --chains-per-gpu $5 ) > $O/leg_$1_$2_compact$c.json 2> $O/leg_$1_$2_compact$c.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/leg_$1_$2_compact$c.json") if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("$1 $2 compact=$c: %.4g steps/s, %.2f s timed, %.2f s warm-up, kernel %s %.3f ms/launch x %d, frac %.3f, slot efficiency %.3f, steady %s, rhat %s" % (
        d["value"], d["seconds_timed"], d["seconds_warmup"], r.get("kernel"), r.get("avg_launch_ms", 0), r.get("launches", 0), r.get("frac", 0),
        r.get("slot_efficiency", 0), (r.get("steady_state") or {}).get("frac"), d.get("rhat_max")))
except Exception as e:
    print("$1 $2 compact=$c: FAILED", e); print(open("$O/leg_$1_$2_compact$c.err").read()[-1500:])
PY
  done
done
( timeout 900 python bench.py --steps 20 --warmup 5 --no-configs --no-inlined --no-cpu-baseline --no-ess ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python -c "
import json; d = json.loads([l for l in open('$O/bench_cfg2.json') if l.startswith('{')][-1]); r = d['roofline']
print('cfg2 headline: %.4g steps/s, %.3f ms/step, %s %.4f ms/launch, frac %.4f' % (d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac']))"
