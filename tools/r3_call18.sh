#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_r; mkdir -p $O; : > $O/diag2.txt
for cfg in "1 8" "2 8" "4 8" "8 8" "8 1" "8 2" "8 4" "4 4" "2 2"; do set -- $cfg
  GU=$1 GK=$2 RH_GRAD_PIPELINE=2 timeout 300 python tools/fuzz_diag.py 1 4096 2>&1 | grep "engine 2 splits 1" >> $O/diag2.txt
done
cat $O/diag2.txt
