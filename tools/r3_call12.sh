#!/bin/bash
# round 3, GPU call 12: the mid-trajectory update as the PROLOGUE of the next gradient launch -- bit-identity, then cfg 2 with and without
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_l; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -x -k "fused or tick_engine or tick" ) > $O/t_a.log 2>&1; tail -6 $O/t_a.log
B="python bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-ess --no-inlined"
for rep in 1 2; do for f in 1 0; do
  RH_FUSE=$f $B > $O/bench_f${f}_$rep.json 2> $O/bench_f${f}_$rep.err
  python - <<PY
import json
try:
  d=json.load(open("$O/bench_f${f}_$rep.json")); r=d["roofline"]; print("RH_FUSE=$f rep $rep: ms_per_step %.3f avg_launch_ms %.4f frac %.4f kernel %s" % (d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["kernel"]))
except Exception as e: print("RH_FUSE=$f failed", e)
PY
done; done
