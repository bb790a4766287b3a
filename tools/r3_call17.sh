#!/bin/bash
# round 3, GPU call 17: the fuzz models through the kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_q; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x ) > $O/t_fuzz.log 2>&1; tail -12 $O/t_fuzz.log
