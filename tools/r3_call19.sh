#!/bin/bash
# round 3, GPU call 19: the whole GPU suite (incl. the fuzz) after the spill guard, then the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_s; mkdir -p $O; rm -f gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/t_all.log 2>&1; tail -6 $O/t_all.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json; tail -4 $O/bench_default.err
cp gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
