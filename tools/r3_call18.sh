#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_r; mkdir -p $O; : > $O/diag4.txt
export RH_KEEP_UNROLL=1 GU=8 GK=8 NP=3 RH_GRAD_PIPELINE=2
echo "--- with sched_barrier" >> $O/diag4.txt
timeout 300 python tools/fuzz_diag.py 1 4096 2>&1 | grep -A1 "fast.*engine 2 splits 1" >> $O/diag4.txt
echo "--- sched_barrier compiled out" >> $O/diag4.txt
RH_HIPRTC_EXTRA='-D__builtin_amdgcn_sched_barrier(x)=' timeout 300 python tools/fuzz_diag.py 1 4096 2>&1 | grep -A1 "fast.*engine 2 splits 1" >> $O/diag4.txt
cat $O/diag4.txt
