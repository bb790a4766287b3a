#!/bin/bash
# round 3, GPU call 23: after the chain-engine row-unroll guard -- the fuzz suite, the wide sweep on BOTH engines (seed 120 included), then as
# much of the GPU suite as the remaining budget allows (heavy-model tests first)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_v; mkdir -p $O
( timeout 60 python -m pytest tests/test_gpu_fuzz.py -q ) > $O/t_fuzz.log 2>&1; tail -2 $O/t_fuzz.log
( timeout 120 python tools/gpu_fuzz_sweep.py run 100 220 ) > $O/sweep_both.txt 2>&1; tail -3 $O/sweep_both.txt
( timeout 170 python -m pytest tests/test_gpu_reference_lowering.py tests/test_gpu_parity.py -q -x -m gpu ) > $O/t_part.log 2>&1; tail -4 $O/t_part.log
