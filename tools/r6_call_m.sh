#!/bin/bash
export RH_DIAG=1
# GPU call M of round 6: the gradient-level slot-independence test -- green on the tree as handed over, and (does it have teeth?) run once more
# with the compiler's own contraction back (RH_XFUSE=0), where it is expected to FAIL on the centred model.  -> gpurun_out/r6_m/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_m; mkdir -p $O
( time RH_HARVEST=$O/kcache_new timeout 600 python -m pytest tests/test_gpu_live_chains.py -m gpu -q --tb=short -rf -p no:cacheprovider --durations=5 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests.log | tail -8
( RH_XFUSE=0 timeout 600 python -m pytest tests/test_gpu_live_chains.py -m gpu -q --tb=line -rf -p no:cacheprovider -k "compacted_list_does_not_depend or compaction_leaves_every_chain_bit_identical or does_not_depend_on_its_neighbours" ) > $O/tests_xfuse0.log 2>&1
echo "--- with RH_XFUSE=0 (expected: failures on case 3):"; grep -E "^FAILED|passed|failed" $O/tests_xfuse0.log | tail -8
