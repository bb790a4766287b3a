#!/bin/bash
# round 3, GPU call 24: the GLM / logistic parity tests after the chain-engine row-unroll guard (their models' kernels were among those re-lowered)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_w; mkdir -p $O
( timeout 110 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "glm or logistic or logit or big_theta or random_walk" ) > $O/t_glm.log 2>&1; tail -3 $O/t_glm.log
