#!/bin/bash
# round 3, GPU call 2: fused-launch synchronisation variants, the table-driven link, the flat-tile gather kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_b; mkdir -p $O; rm -f gpurun_out/baseline_samplers.txt
B="python bench.py --no-cpu-baseline --no-ess --no-inlined --steps 40 --warmup 20"
for v in 2 1; do ( RH_FUSE_SYNC=$v timeout 200 $B ) > $O/bench_sync$v.json 2> $O/bench_sync$v.err; python - <<PY
import json
d=json.loads(open("$O/bench_sync$v.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("sync $v: ms_per_step %.3f avg_launch_ms %.4f all_kernels_ms/step %.3f kernel %s" % (d["ms_per_step"], r["avg_launch_ms"], r["all_kernels_ms"]/d["steps"], r["kernel"]))
PY
done
( RH_FUSE=0 timeout 200 $B ) > $O/bench_unfused.json 2> $O/bench_unfused.err; python -c "
import json
d=json.loads(open('$O/bench_unfused.json').read().strip().splitlines()[-1]); r=d['roofline']
print('unfused: ms_per_step %.3f avg_launch_ms %.4f' % (d['ms_per_step'], r['avg_launch_ms']))"
( time timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -x -k "fused or tick_engine or logit or negbin or gather or big_mode or big_table or glm or cfg2_full" ) > $O/t_a.log 2>&1; tail -6 $O/t_a.log
( RH_FUSE_SYNC=1 timeout 300 python -m pytest tests/test_gpu_fused.py -q -x ) > $O/t_sync1.log 2>&1; tail -3 $O/t_sync1.log
( timeout 300 python tools/cfg5_probe.py 10000 100 1024 ) > $O/cfg5_probe.txt 2>&1; tail -4 $O/cfg5_probe.txt
( timeout 300 python tools/cfg4_probe.py 10000000 256 2 ) > $O/cfg4_probe.txt 2>&1; tail -2 $O/cfg4_probe.txt
( time timeout 600 python -m pytest tests/test_gpu_baseline_samplers.py -q -s -k "cfg4 or cfg5" ) > $O/t_samplers.log 2>&1; tail -5 $O/t_samplers.log; cat gpurun_out/baseline_samplers.txt
