#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# GPU call D of round 6 (second session): the whole -m gpu tier against the driver's 1200 s limit, the compacted-vs-every-chain
# discrepancy on the centred model taken apart, what GLMMPoisson2 costs under the dynamic samplers, the default bench line under
# rocprofv3 --kernel-trace --stats.  -> gpurun_out/r6_d/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_d; mkdir -p $O
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 1700 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider --durations=30 --timeout 420 \
    -k "not glmm_poisson2 or not nuts_posterior" ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests.log | tail -30
( timeout 700 python tools/r6_live_diag.py 700 100 ) > $O/live_diag.txt 2>&1; cut -c1-2500 $O/live_diag.txt
( timeout 420 python tools/r6_glmm_timing.py ) > $O/glmm_timing.txt 2>&1; cat $O/glmm_timing.txt | cut -c1-400
unset RH_DIAG
( timeout 900 bash tools/pmc_cfg2.sh ${1:-unknown} r6_cfg2 ) 2>&1 | tail -12
