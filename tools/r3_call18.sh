#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_r; mkdir -p $O; : > $O/diag3.txt
for p in 2 0; do NP=11 RH_GRAD_PIPELINE=$p timeout 300 python tools/fuzz_diag.py 11 70 2>&1 | tail -16 >> $O/diag3.txt; done
NP=3 RH_GRAD_PIPELINE=2 timeout 300 python tools/fuzz_diag.py 11 70 2>&1 | tail -16 >> $O/diag3.txt
cat $O/diag3.txt
