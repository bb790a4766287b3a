#!/bin/bash
# round 3, GPU call 3: fused launch with prefetch + interleaved reductions, LDS link table, two-accumulator gather walk, big-mode tick
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_c; mkdir -p $O; rm -f gpurun_out/baseline_samplers.txt
B="python bench.py --no-cpu-baseline --no-ess --no-inlined --steps 40 --warmup 20"
for tag in fused unfused; do
  if [ $tag = unfused ]; then export RH_FUSE=0; else unset RH_FUSE; fi
  ( timeout 200 $B ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag: ms_per_step %.3f avg_launch_ms %.4f all_kernels_ms/step %.3f kernel %s frac %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['all_kernels_ms']/d['steps'], r['kernel'], r['frac']))"
done; unset RH_FUSE
( time timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -x -k "fused or tick_engine or logit or negbin or gather or big_mode or big_table or glm or cfg2_full or nuts" ) > $O/t_a.log 2>&1; tail -6 $O/t_a.log
( RH_GATHER_SCAN=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gather or big_table or negbin" ) > $O/t_scan.log 2>&1; tail -3 $O/t_scan.log
( timeout 300 python tools/cfg5_probe.py 10000 100 1024 ) > $O/cfg5_probe.txt 2>&1; tail -4 $O/cfg5_probe.txt
( RH_GATHER_SCAN=1 timeout 300 python tools/cfg5_probe.py 10000 100 1024 ) > $O/cfg5_probe_scan.txt 2>&1; tail -1 $O/cfg5_probe_scan.txt
( timeout 300 python tools/cfg4_probe.py 10000000 256 2 ) > $O/cfg4_probe.txt 2>&1; tail -1 $O/cfg4_probe.txt
( RH_LK_LDS=0 timeout 300 python tools/cfg4_probe.py 10000000 256 2 ) > $O/cfg4_probe_glob.txt 2>&1; tail -1 $O/cfg4_probe_glob.txt
( time timeout 600 python -m pytest tests/test_gpu_baseline_samplers.py -q -s -k "cfg5" ) > $O/t_samplers.log 2>&1; tail -3 $O/t_samplers.log; cat gpurun_out/baseline_samplers.txt
