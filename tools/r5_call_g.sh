#!/bin/bash
# GPU call G of round 5: the wide fuzz sweep, seeds 332..380 (64 cases, two builds each) against the oracle.  -> gpurun_out/r5_g/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_g; mkdir -p $O
( time timeout 1200 python tools/gpu_fuzz_sweep.py run 332 380 ) > $O/fuzz_sweep_332_380.txt 2>&1
grep -E "FAIL|ERROR|sweep" $O/fuzz_sweep_332_380.txt | cut -c1-300 | tail -12; tail -4 $O/fuzz_sweep_332_380.txt
