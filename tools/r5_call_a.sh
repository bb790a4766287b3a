#!/bin/bash
# GPU call A of round 5: (1) the new group-major gather walk + the strict-gather lowering against the oracle (the tests that touch
# them), (2) cfg 5 probe variants (walk, K, wavefronts per SIMD, row splits), (3) counters for the old and the new walk,
# (4) cfg 4 baseline + counters.  Everything lands under gpurun_out/r5_a/.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_a; mkdir -p $O
( time timeout -s INT 1100 python -m pytest tests/test_gpu_strict_gather.py tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_reference_lowering.py tests/test_gpu_baseline_samplers.py \
    -m gpu -v --tb=short -rf -p no:cacheprovider --durations=15 \
    -k "gather or big_mode or cfg5 or hierarchical or strict_glmm or strict_cfg5 or strict_location or strict_table or strict_reference" ) > $O/tests.log 2>&1
grep -E "FAILED|ERROR|passed|failed" $O/tests.log | tail -15
cp gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
i=0
for v in "K=4" "K=4 RH_GATHER_V2=0" "K=2 RH_GATHER_WAVES=4" "K=4 RH_GATHER_WAVES=1" "K=2"; do
  i=$((i+1)); K=$(echo $v | sed 's/K=\([0-9]*\).*/\1/'); E=$(echo $v | sed 's/K=[0-9]* *//')
  ( env $E RH_PROBE_SPLITS=0,8,16,24 timeout 300 python tools/cfg5_probe.py 10000 100 1024 $K ) > $O/probe_$i.txt 2>&1
  echo "== $v"; grep '^{"G"' $O/probe_$i.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   splits %2d: gather %.3f ms, per step %.3f ms' % (d['splits'], d['grad_kernel_ms'], d['all_ms']))"
done
cd /tmp && export TMPDIR=/tmp
for v in "v2:" "old:RH_GATHER_V2=0"; do
  name=${v%%:*}; E=${v#*:}; j=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    j=$((j+1)); mkdir -p $O/pmc_$name/p$j
    env $E rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_$name/p$j -o bench -- python $GRAFT_REPO_ROOT/tools/cfg5_probe.py 10000 100 1024 4 > $O/pmc_$name/p$j/log.txt 2>&1
    f=$(find $O/pmc_$name/p$j -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_$name/p$j/bench_counter_collection.csv
  done
  python $GRAFT_REPO_ROOT/profiles/summarize.py rh_grad_gather_kernel 24 $O/pmc_gather_$name.json $O/pmc_$name/p1 $O/pmc_$name/p2 $O/pmc_$name/p3 $O/pmc_$name/p4 $O/pmc_$name/p5 $O/pmc_$name/p6 > $O/pmc_gather_$name.txt 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py rh_tick_kernel 24 $O/pmc_tick_$name.json $O/pmc_$name/p1 $O/pmc_$name/p2 $O/pmc_$name/p4 > /dev/null 2>&1
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/pmc_$name/stats -o bench -- python $GRAFT_REPO_ROOT/tools/cfg5_probe.py 10000 100 1024 4 > $O/pmc_$name/stats_log.txt 2>&1
  f=$(find $O/pmc_$name/stats -name "bench_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$name.csv
  rm -rf $O/pmc_$name
  echo "== counters $name"; python -c "
import json; d = json.load(open('$O/pmc_gather_$name.json'))['counters']
print({k: (round(v['mean_per_launch']), round(v['mean_duration_us'])) for k, v in d.items()})"
done
# cfg 4: where it stands (2e6 rows x 256 chains probe as in round 3 / 4) + its counters
( timeout 300 python $GRAFT_REPO_ROOT/tools/cfg4_probe.py 2000000 256 2 ) > $O/cfg4_probe.txt 2>&1; tail -1 $O/cfg4_probe.txt
RH_PMC_KERNEL=rh_grad_glm_kernel bash $GRAFT_REPO_ROOT/tools/pmc_cfg4.sh > $O/cfg4_pmc.txt 2>&1
cp $GRAFT_REPO_ROOT/gpurun_out/pmc_glm.json $O/cfg4_pmc_glm.json 2>/dev/null; rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_cfg4
python -c "
import json; d = json.load(open('$O/cfg4_pmc_glm.json'))['counters']
print({k: (round(v['mean_per_launch']), round(v['mean_duration_us'])) for k, v in d.items()})"
