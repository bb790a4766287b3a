#!/bin/bash
# round 3, GPU call 13: the whole GPU suite at the new defaults (rolling row loop, prologue-fused launches), then the evidence for
# the bench command: kernel stats + PMC passes (tools/pmc_cfg2.sh) and the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_m; mkdir -p $O; rm -f gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt
HEAD=${1:-unknown}
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/t_all.log 2>&1; tail -8 $O/t_all.log
bash tools/pmc_cfg2.sh $HEAD r3_b_cfg2 > $O/pmc_cfg2.log 2>&1; tail -12 $O/pmc_cfg2.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/bench_default.json; tail -4 $O/bench_default.err
cp gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
