#!/bin/bash
# GPU call B of round 5: (1) cfg 5 -- do chains with the same parameters get the same bits (the check that failed in call A);
# gather walk with the select-masked ragged tile, the fused big-mode tick, split counts; (2) cfg 4 -- rh_grad_glm4r_kernel against
# rh_grad_glm_kernel: the GLM parity tests on it, then the 2e6-row probe and counters.  -> gpurun_out/r5_b/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_b; mkdir -p $O
echo "== cfg5 identical-chains diagnostic"
( timeout 200 python tools/cfg5_diag.py 1024 0 ) > $O/diag_default.txt 2>&1; grep -E "vector|Error|error" $O/diag_default.txt | cut -c1-400
( RH_GATHER_TAIL_SELECT=0 timeout 200 python tools/cfg5_diag.py 1024 0 ) > $O/diag_tail_exec.txt 2>&1; grep -E "vector|Error|error" $O/diag_tail_exec.txt | cut -c1-400
echo "== cfg5 probes"
i=0
for v in "K=0" "K=0 RH_TICK_FAST=0" "K=0 RH_GATHER_V2=0"; do
  i=$((i+1)); K=$(echo $v | sed 's/K=\([0-9]*\).*/\1/'); E=$(echo $v | sed 's/K=[0-9]* *//')
  ( env $E RH_PROBE_SPLITS=0,16,24,32,48 timeout 300 python tools/cfg5_probe.py 10000 100 1024 $K ) > $O/probe_$i.txt 2>&1
  echo "-- $v"; grep '^{"G"' $O/probe_$i.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   splits %2d: gather %.3f ms, per step %.3f ms' % (d['splits'], d['grad_kernel_ms'], d['all_ms']))"
done
echo "== GLM parity tests on rh_grad_glm4r_kernel"
( time RH_GLM4R=1 timeout -s INT 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_reference_lowering.py -m gpu -q --tb=short -rf -p no:cacheprovider --durations=8 \
    -k "glm or logistic or cfg4 or more_than_128" ) > $O/tests_glm4r.log 2>&1
grep -E "FAILED|ERROR|passed|failed" $O/tests_glm4r.log | tail -12
echo "== cfg4 probes (2e6 rows x 256 chains)"
for v in "RH_GLM4R=0" "RH_GLM4R=1"; do
  ( env $v timeout 300 python tools/cfg4_probe.py 2000000 256 2 ) > $O/cfg4_probe_${v#*=}.txt 2>&1; echo "-- $v"; tail -1 $O/cfg4_probe_${v#*=}.txt | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$O/pmc4r; i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM FETCH_SIZE"; do
  i=$((i+1)); mkdir -p $P/p$i
  RH_GLM4R=1 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $P/p$i -o bench -- python $R/tools/cfg4_probe.py 2000000 256 2 > $P/p$i/log.txt 2>&1
  f=$(find $P/p$i -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$P/p$i/bench_counter_collection.csv" ] && cp $f $P/p$i/bench_counter_collection.csv
done
python $R/profiles/summarize.py rh_grad_glm4r_kernel 8 $O/pmc_glm4r.json $P/p1 $P/p2 $P/p3 $P/p4 $P/p5 > $O/pmc_glm4r.txt 2>&1
rm -rf $P
python -c "
import json; d = json.load(open('$O/pmc_glm4r.json'))['counters']
print({k: (round(v['mean_per_launch']), round(v['mean_duration_us'])) for k, v in d.items()})"
