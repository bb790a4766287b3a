#!/bin/bash
export RH_DIAG=1
# cfg 4 (logistic 1e7 x 50, 256 chains, static HMC(8)): rh_grad_glm_kernel's four evaluations per lane as ONE block per variant (default) against
# the first round-6 form with the gradient-only choice inside the unrolled loop (RH_GLM_ELEM_INLOOP=1), each with and without gradient-only
# launches.  Same box, back to back.  -> gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_cfg4}; mkdir -p $O
for form in 0 1; do
  for vf in 1 0; do
    extra=""; [ $form = 1 ] && extra="-DRH_GLM_ELEM_INLOOP=1"
    ( RH_HIPRTC_EXTRA="$extra" RH_VALUE_FREE=$vf timeout 400 python bench.py --workload cfg4 --sampler hmc8 --steps 3 --warmup 2 --chains-per-gpu 256 ) > $O/ab_inloop${form}_vf${vf}.json 2> $O/ab_inloop${form}_vf${vf}.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/ab_inloop${form}_vf${vf}.json") if l.startswith("{")][-1]); r = d["roofline"]
    print("cfg4 hmc8 INLOOP=$form VALUE_FREE=$vf: %.3f ms/launch x %d, frac %.3f" % (r["avg_launch_ms"], r["launches"], r["frac"]))
except Exception as e:
    print("cfg4 INLOOP=$form VALUE_FREE=$vf FAILED", e); print(open("$O/ab_inloop${form}_vf${vf}.err").read()[-800:])
PY
  done
done
