#!/bin/bash
# round 3, GPU call 7: the two shortened sampler tests, side numbers for DESIGN 3.7 (strict cfg 2, cfg 3 EHMC / NUTS, cfg 5 NUTS at 1024 chains)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_g; mkdir -p $O; rm -f gpurun_out/baseline_samplers.txt
( time timeout 900 python -m pytest tests/test_gpu_baseline_samplers.py -q -s -k "cfg4 or centred" ) > $O/t_s.log 2>&1; tail -4 $O/t_s.log; cat gpurun_out/baseline_samplers.txt
echo "strict cfg2: $(timeout 300 python bench.py --strict --no-cpu-baseline --no-ess --no-inlined --steps 10 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.2f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))")"
for a in "--workload cfg3 --steps 100 --warmup 200" "--workload cfg3 --sampler nuts --steps 100 --warmup 200" "--workload cfg3 --steps 100 --warmup 200 --chains-per-gpu 32768" "--workload cfg1 --steps 400 --warmup 100 --chains-per-gpu 32768"; do
  echo "$a: $(timeout 300 python bench.py $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e steps/s, ess/s %s, mean leapfrog %.1f' % (d['value'], d.get('ess_per_s'), d.get('mean_leapfrog_per_iteration', 0)))")"
done
echo "cfg5 NUTS 1024 chains: $(timeout 600 python tools/cfg5_probe.py 10000 100 1024 0 3 2>&1 | tail -1)"
