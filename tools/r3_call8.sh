#!/bin/bash
# round 3, GPU call 8: rolling row-loop pipeline (RH_GRAD_PIPELINE=2) against the default loop on cfg 2, same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_h; mkdir -p $O
B="python bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-ess --no-inlined"
for rep in 1 2; do
for v in 0 2; do
  RH_GRAD_PIPELINE=$v $B > $O/bench_p${v}_$rep.json 2> $O/bench_p${v}_$rep.err
  python - <<PY
import json
try:
  d=json.load(open("$O/bench_p${v}_$rep.json")); print("pipeline $v rep $rep: ms_per_step %.3f avg_launch_ms %.4f frac %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e: print("pipeline $v failed", e)
PY
done; done
for u in 2 8; do
  RH_GRAD_PIPELINE=2 $B --grad-unroll $u > $O/bench_p2_u$u.json 2> $O/bench_p2_u$u.err
  python - <<PY
import json
try:
  d=json.load(open("$O/bench_p2_u$u.json")); print("pipeline 2 unroll $u: ms_per_step %.3f avg_launch_ms %.4f frac %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e: print("pipeline 2 unroll $u failed", e)
PY
done
( RH_GRAD_PIPELINE=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -m gpu -q -x ) > $O/t_p2.log 2>&1; tail -4 $O/t_p2.log
