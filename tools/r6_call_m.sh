#!/bin/bash
export RH_DIAG=1
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_m; mkdir -p $O
( time RH_HARVEST=$O/kcache_new timeout 600 python -m pytest tests/test_gpu_live_chains.py -m gpu -q --tb=short -rf -p no:cacheprovider --durations=5 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests.log | tail -8; ls $O/kcache_new 2>/dev/null | wc -l
