#!/bin/bash
# GPU call J of round 6, the tree as handed over: the -m gpu tier once more (bench.py changed since call F), the driver's bench command with the
# new configs plan, and a wide fuzz sweep on fresh seeds through the explicit-fusion row code (480..519, strict + fast builds, both engines,
# sampler kernels against the oracle's chains).  -> gpurun_out/r6_j/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_j; mkdir -p $O
( time RH_DIAG=1 RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 1200 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider --durations=12 --timeout 420 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^real" $O/tests.log | tail -12
bash tools/r6_call_i.sh 2>&1 | tail -14; mv gpurun_out/r6_i/* $O/ 2>/dev/null
( time SWEEP_CHAINS=1 timeout 600 python tools/gpu_fuzz_sweep.py run 480 520 ) > $O/fuzz_480_520.txt 2>&1; grep -E "^FAIL|^ERROR|^sweep|^real" $O/fuzz_480_520.txt | tail -8
