#!/bin/bash
# round 3, GPU call 15: roofline.traffic measured inside the bench run (rocprofv3 --pmc child runs); the torchrun bench test; strict cfg 2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_o; mkdir -p $O
( time timeout 600 python bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-ess --no-inlined ) > $O/bench_live.json 2> $O/bench_live.err
python - <<PY
import json
d=json.load(open("$O/bench_live.json")); r=d["roofline"]
print("ms_per_step %.3f frac %.4f traffic %s\n  source: %s %s" % (d["ms_per_step"], r["frac"], r["traffic"], r["traffic_source"], r.get("live_traffic_failed","")))
PY
tail -3 $O/bench_live.err
( time timeout 900 python -m pytest tests/test_gpu_multi.py -q -x -k "torch or bench or driver" ) > $O/t_multi.log 2>&1; tail -4 $O/t_multi.log
( timeout 300 python bench.py --strict --steps 8 --warmup 8 --no-cpu-baseline --no-ess --no-inlined --no-live-traffic ) > $O/bench_strict.json 2> $O/bench_strict.err
python -c "
import json; d=json.load(open('$O/bench_strict.json')); print('strict cfg2: ms_per_step %.2f frac %.4f kernel %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))"
