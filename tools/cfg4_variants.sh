#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# cfg-4-shaped probe (rh_grad_glm_kernel) under the knobs that shape the kernel; the code objects are precompiled on the
# build host (tools/cfg4_variants.sh precompile) so that the GPU box only loads them.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
VARIANTS=("" "RH_LOGIT_LINK=0" "RH_GLM_WPS=1" "RH_GLM_WPS=3" "RH_GLM_W=4" "RH_GLM_W=16" "RH_GLM_EU=1" "RH_GLM_EU=2")
if [ "$1" = "precompile" ]; then
  for v in "${VARIANTS[@]}"; do
    env $v python - <<'PY'
from rainier_amd import _capi, models
src, size = _capi.lower_only(models.logistic(n=8, k=50).rir, _capi.compile_opts(fp_contract=True, factor_outputs=True))
print(size)
PY
  done
  exit 0
fi
ROWS=${1:-2000000}
for v in "${VARIANTS[@]}"; do
  echo "== variant [$v] rows $ROWS"
  env $v python $R/tools/cfg4_probe.py $ROWS 256 2 2>&1 | tail -1
done
