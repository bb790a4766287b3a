#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# GPU call C of round 6: where the centred hierarchical model's chains pick up a dependency on their launch companions (diagnostic), the
# tests added since call B, the NUTS legs call B did not reach, gradient-only vs full launches.  -> gpurun_out/r6_c/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_c; mkdir -p $O
( timeout 900 python tools/r6_centred_diag.py 700 100 ) > $O/centred_diag.txt 2>&1; cat $O/centred_diag.txt | cut -c1-1500
( time timeout -s INT 1500 python -m pytest tests/test_gpu_reference_lowering.py tests/test_gpu_nuts_distribution.py tests/test_gpu_live_chains.py tests/test_gpu_baseline_sizes.py tests/test_gpu_parity.py \
    -m gpu -q --tb=short -rf -p no:cacheprovider --durations=8 \
    -k "second_data_set or nuts_posterior or gradient_only or fast_path_is_bit or headline_kernel or logit_link or compacted_nuts" ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests.log | tail -12
unset RH_DIAG
for leg in "cfg5c default 60 36 1024" "cfg4 default 100 60 256" "cfg5 hmc8 4 2 1024" "cfg4 hmc8 2 2 256"; do
  set -- $leg
  ( timeout 600 python bench.py --workload $1 --sampler $2 --steps $3 --warmup $4 --chains-per-gpu $5 ) > $O/leg_$1_$2.json 2> $O/leg_$1_$2.err
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
export RH_DIAG=1
for w in "cfg5 1024 4" "cfg4 256 2"; do
  set -- $w
  for v in 1 0; do
    ( RH_VALUE_FREE=$v timeout 600 python bench.py --workload $1 --sampler hmc8 --steps $3 --warmup 2 --chains-per-gpu $2 ) > $O/vf_$1_$v.json 2>/dev/null
    python -c "
import json; d = json.loads([l for l in open('$O/vf_$1_$v.json') if l.startswith('{')][-1]); r = d['roofline']
print('$1 hmc8 RH_VALUE_FREE=$v: %.3f ms/launch, frac %.3f' % (r['avg_launch_ms'], r['frac']))"
  done
done
