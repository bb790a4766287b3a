#!/bin/bash
# round 3, GPU call 11: pipelined gather walk -- parity of every gather-mode test, cfg 5 probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_k; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -x -k "negbin or gather or big_table or big_mode or cfg5 or centred or hier" ) > $O/t_a.log 2>&1; tail -6 $O/t_a.log
( timeout 300 python tools/cfg5_probe.py 10000 100 1024 ) > $O/cfg5_probe.txt 2>&1; tail -4 $O/cfg5_probe.txt
