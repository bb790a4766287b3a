#!/bin/bash
export RH_DIAG=1
# GPU call H of round 6: (1) the row-split multiplier of the dynamic samplers (RH_SUBF; 8 by default while a split keeps >= 1024 rows) on cfg 2
# under DefaultConfig -- fewer, longer splits cost less per launch when every chain is live, more when few are; (2) cfg 5 (centred) with
# DefaultConfig's own mass windows (warm-up 150), as call G found for cfg 4 (mean tree 90 -> 9 leapfrog steps).  -> gpurun_out/r6_h/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_h; mkdir -p $O
show() {  # file label
  python - "$1" "$2" <<'PY'
import json, sys
f, label = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d.get("roofline") or {}
    print("%s: %.4g steps/s, warm-up %.1f s, timed %.1f s, %.3f ms/launch x %d, frac %.3f (steady %s), slot eff %s, rhat %.4f, ess/s %s, mean L %.1f" % (
        label, d["value"], d["seconds_warmup"], d["seconds_timed"], r.get("avg_launch_ms", 0), r.get("launches", 0), r.get("frac", 0), (r.get("steady_state") or {}).get("frac"),
        r.get("slot_efficiency"), d.get("rhat_max") or -1, d.get("ess_per_s"), d.get("mean_leapfrog_per_iteration", 0)))
except Exception as e:
    print(label, "FAILED", e); print(open(f[:-5] + ".err").read()[-500:])
PY
}
for sf in 8 4 2 1; do
  ( RH_SUBF=$sf timeout 300 python bench.py --workload cfg2d --sampler default --steps 256 --warmup 200 --chains-per-gpu 1024 ) > $O/cfg2d_subf$sf.json 2> $O/cfg2d_subf$sf.err
  show $O/cfg2d_subf$sf.json "cfg2d RH_SUBF=$sf"
done
unset RH_DIAG
( time timeout 580 python bench.py --workload cfg5c --sampler default --steps 40 --warmup 150 --chains-per-gpu 1024 ) > $O/cfg5c_w150.json 2> $O/cfg5c_w150.err
show $O/cfg5c_w150.json "cfg5c NUTS warm-up 150"
