#!/bin/bash
# round 3, GPU call 9: where the gain of call 8's (rolling pipeline, unroll 8) line comes from -- same box, cfg 2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_i; mkdir -p $O
B="python bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-ess --no-inlined"
run() { # name, env..., -- args
  name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" $B "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
  d=json.load(open("$O/$name.json")); print("$name: ms_per_step %.3f avg_launch_ms %.4f frac %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e: print("$name failed", e)
PY
}
run p0_u4 RH_GRAD_PIPELINE=0 --
run p0_u8 RH_GRAD_PIPELINE=0 -- --grad-unroll 8
run p1_u8 RH_GRAD_PIPELINE=1 -- --grad-unroll 8
run p2_u8 RH_GRAD_PIPELINE=2 -- --grad-unroll 8
run p2_u6 RH_GRAD_PIPELINE=2 -- --grad-unroll 6
run p2_u12 RH_GRAD_PIPELINE=2 -- --grad-unroll 12
run p2_u8_s24 RH_GRAD_PIPELINE=2 -- --grad-unroll 8 --grad-splits 24
run p2_u8_s16 RH_GRAD_PIPELINE=2 -- --grad-unroll 8 --grad-splits 16
run p2_u8_s48 RH_GRAD_PIPELINE=2 -- --grad-unroll 8 --grad-splits 48
run p2_u8_k4 RH_GRAD_PIPELINE=2 -- --grad-unroll 8 --grad-chains 4
run p2_u8_again RH_GRAD_PIPELINE=2 -- --grad-unroll 8
