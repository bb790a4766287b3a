#!/bin/bash
# GPU call C of round 5: the whole GPU tier at the current defaults, the GLM tests again on rh_grad_glm4r_kernel (RH_GLM4R=1),
# cfg 4 / cfg 5 probes + counters, smoke(), the default bench line (with its `configs` block).  -> gpurun_out/r5_c/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_c; mkdir -p $O
ls rainier_amd/kcache > $O/kcache_before.txt
( time RH_HARVEST=$O/kcache_new timeout -s INT --kill-after=60 1100 python -m pytest tests -m gpu -v --tb=short -rf -p no:cacheprovider --durations=25 ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests.log | tail -25
cp gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
echo "== GLM tests on rh_grad_glm4r_kernel (JG = 2)"
( time RH_GLM4R=1 timeout -s INT 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_reference_lowering.py -m gpu -q --tb=short -rf -p no:cacheprovider \
    -k "glm or logistic or cfg4 or more_than_128" ) > $O/tests_glm4r.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests_glm4r.log | tail -12
echo "== cfg4 probes (2e6 rows x 256 chains)"
i=0
for v in "RH_GLM4R=0" "RH_GLM4R=1" "RH_GLM4R=1 RH_GLM4R_W=8" "RH_GLM4R=1 RH_GLM4R_W=2" "RH_GLM4R=1 RH_GLM4R_JG=4"; do
  i=$((i+1)); ( env $v timeout 200 python tools/cfg4_probe.py 2000000 256 2 ) > $O/cfg4_probe_$i.txt 2>&1; echo "-- $v"; tail -1 $O/cfg4_probe_$i.txt | cut -c1-260
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$O/pmc4r; i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM FETCH_SIZE"; do
  i=$((i+1)); mkdir -p $P/p$i
  RH_GLM4R=1 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $P/p$i -o bench -- python $R/tools/cfg4_probe.py 2000000 256 2 > $P/p$i/log.txt 2>&1
  f=$(find $P/p$i -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$P/p$i/bench_counter_collection.csv" ] && cp $f $P/p$i/bench_counter_collection.csv
done
python $R/profiles/summarize.py rh_grad_glm4r_kernel 8 $O/pmc_glm4r_jg2.json $P/p1 $P/p2 $P/p3 $P/p4 $P/p5 > $O/pmc_glm4r_jg2.txt 2>&1
rm -rf $P
python -c "
import json; d = json.load(open('$O/pmc_glm4r_jg2.json'))['counters']
print({k: (round(v['mean_per_launch']), round(v['mean_duration_us'])) for k, v in d.items()})"
echo "== cfg5 probe + kernel stats"
cd $R
( RH_PROBE_SPLITS=0,32,64,96 timeout 300 python tools/cfg5_probe.py 10000 100 1024 0 ) > $O/cfg5_probe.txt 2>&1
grep '^{"G"' $O/cfg5_probe.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   splits %2d: gather %.3f ms, per step %.3f ms' % (d['splits'], d['grad_kernel_ms'], d['all_ms']))"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -o bench -- python $R/tools/cfg5_probe.py 10000 100 1024 0 > $O/cfg5_stats_log.txt 2>&1
f=$(find $O/stats5 -name "bench_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg5_kernel_stats.csv; rm -rf $O/stats5; head -4 $O/cfg5_kernel_stats.csv
cd $R
echo "== smoke + bench"
( time timeout 120 python -c "import __graft_entry__ as G; G.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log | head -2
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    for k, v in d.get("configs", {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "seconds_total", "error", "skipped")}, (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("avg_launch_ms"))
except Exception as e:
    print("bench output unreadable:", e)
PY
tail -3 $O/bench.err
# what this call compiled (models build() / the dry lowering did not know): back into the in-tree kernel cache
mkdir -p $O/kcache_new; ls rainier_amd/kcache | sort > $O/kcache_after.txt
comm -13 <(sort $O/kcache_before.txt) $O/kcache_after.txt | grep -v "\.tmp" | while read f; do cp -n rainier_amd/kcache/$f $O/kcache_new/ 2>/dev/null; done
ls $O/kcache_new | wc -l; du -sh $O/kcache_new | cut -f1
