#!/bin/bash
# round 3, GPU call 20: the fuzz, then the whole GPU suite after the spill guard's second half
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_t; mkdir -p $O; rm -f gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt
( time timeout 600 python -m pytest tests/test_gpu_fuzz.py -q ) > $O/t_fuzz.log 2>&1; tail -6 $O/t_fuzz.log
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fuzz.py ) > $O/t_all.log 2>&1; tail -6 $O/t_all.log
cp gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
