#!/bin/bash
# round 3, GPU call 16: rh_grad_glmv_kernel (one chain per lane, row values in scalar registers) against the MFMA GLM kernel on cfg 4
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_p; mkdir -p $O
( RH_GLMV=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "glm or logistic" ) > $O/t_glmv.log 2>&1; tail -4 $O/t_glmv.log
for v in 1 0; do
  ( RH_GLMV=$v timeout 300 python tools/cfg4_probe.py 10000000 256 2 ) > $O/cfg4_glmv$v.txt 2>&1; echo "RH_GLMV=$v: $(tail -1 $O/cfg4_glmv$v.txt)"
done
( RH_GLMV=1 timeout 300 python tools/cfg4_probe.py 2000000 1024 2 ) > $O/cfg4_glmv1_1024.txt 2>&1; echo "RH_GLMV=1 2e6 x 1024: $(tail -1 $O/cfg4_glmv1_1024.txt)"
( RH_GLMV=0 timeout 300 python tools/cfg4_probe.py 2000000 1024 2 ) > $O/cfg4_glmv0_1024.txt 2>&1; echo "RH_GLMV=0 2e6 x 1024: $(tail -1 $O/cfg4_glmv0_1024.txt)"
