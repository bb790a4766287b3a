#!/bin/bash
# GPU call P of round 6: the driver's own round-end sequence on the tree as handed over -- `pytest tests -x -q -m gpu`, then smoke().  -> gpurun_out/r6_p/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_p; mkdir -p $O
( time timeout 1200 python -m pytest tests/ -x -q -m gpu ) > $O/tests.log 2>&1; tail -6 $O/tests.log
( time timeout 120 python -c "import __graft_entry__ as G; G.smoke()" ) > $O/smoke.log 2>&1; head -2 $O/smoke.log
