#!/bin/bash
# round 3, GPU call 14: the default bench line at HEAD (roofline.traffic from profiles/r3_b_cfg2, same generated source)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_n; mkdir -p $O
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json; tail -4 $O/bench_default.err
