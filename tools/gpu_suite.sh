#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# The GPU tier in one call: every -m gpu test (no -x: a failure must not hide the rest; the row representatives of tests/conftest.py
# run first), smoke(), and the default bench line.  usage, from the repo root:
#   gpurun --timeout 1500 -- "bash tools/gpu_suite.sh NAME [pytest args]"      -> gpurun_out/NAME/{tests.log,smoke.log,bench.json}
cd "$GRAFT_REPO_ROOT" || exit 1
NAME=${1:-suite}; shift
O=gpurun_out/$NAME; mkdir -p "$O"
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 ${SUITE_TIMEOUT:-1200} python -m pytest tests -m gpu -v --tb=short -rf -p no:cacheprovider --durations=25 "$@" ) > "$O/tests.log" 2>&1
grep -E "FAILED|ERROR|passed|failed" "$O/tests.log" | tail -30
[ -n "$SUITE_ONLY" ] && exit 0
( time timeout 120 python -c "import __graft_entry__ as G; G.smoke()" ) > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log"
( time timeout 400 python bench.py ) > "$O/bench.json" 2> "$O/bench.err"; cut -c1-600 "$O/bench.json"; tail -3 "$O/bench.err"
