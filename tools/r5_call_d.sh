#!/bin/bash
# GPU call D of round 5: rh_grad_glm4r_kernel with hand-phased operand reads (probe + the GLM tests on it), the three tests call C
# left red, the default bench line with the static-HMC legs in its `configs` block.  -> gpurun_out/r5_d/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_d; mkdir -p $O
ls rainier_amd/kcache | sort > $O/kcache_before.txt
echo "== cfg4 probes (2e6 rows x 256 chains)"
i=0
for v in "RH_GLM4R=0" "RH_GLM4R=1"; do
  i=$((i+1)); ( env $v timeout 200 python tools/cfg4_probe.py 2000000 256 2 ) > $O/cfg4_probe_$i.txt 2>&1; echo "-- $v"; tail -1 $O/cfg4_probe_$i.txt | cut -c1-260
done
echo "== GLM tests on rh_grad_glm4r_kernel"
( time RH_GLM4R=1 timeout -s INT 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_reference_lowering.py -m gpu -q --tb=short -rf -p no:cacheprovider \
    -k "glm or logistic or cfg4 or more_than_128" ) > $O/tests_glm4r.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests_glm4r.log | tail -8
echo "== the tests call C left red + the clone test"
( time timeout -s INT 600 python -m pytest tests/test_jni_shim.py tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q --tb=short -rf -rs -p no:cacheprovider \
    -k "shim or lds_staged or clone_takes" ) > $O/tests_fixed.log 2>&1
grep -E "^FAILED|^ERROR|^SKIPPED|passed|failed" $O/tests_fixed.log | cut -c1-300 | tail -8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$O/pmc4r; i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1)); mkdir -p $P/p$i
  RH_GLM4R=1 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $P/p$i -o bench -- python $R/tools/cfg4_probe.py 2000000 256 2 > $P/p$i/log.txt 2>&1
  f=$(find $P/p$i -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$P/p$i/bench_counter_collection.csv" ] && cp $f $P/p$i/bench_counter_collection.csv
done
python $R/profiles/summarize.py rh_grad_glm4r_kernel 8 $O/pmc_glm4r_phased.json $P/p1 $P/p2 $P/p3 > $O/pmc_glm4r_phased.txt 2>&1
rm -rf $P
python -c "
import json; d = json.load(open('$O/pmc_glm4r_phased.json'))['counters']
print({k: (round(v['mean_per_launch']), round(v['mean_duration_us'])) for k, v in d.items()})"
cd $R
echo "== bench"
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    for k, v in d.get("configs", {}).items():
        r = v.get("roofline") or {}
        print("  ", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "mean_leapfrog_per_iteration", "seconds_total", "error", "skipped")}, r.get("kernel"), r.get("frac"), r.get("avg_launch_ms"))
except Exception as e:
    print("bench output unreadable:", e)
PY
tail -3 $O/bench.err
mkdir -p $O/kcache_new; ls rainier_amd/kcache | sort > $O/kcache_after.txt
comm -13 $O/kcache_before.txt $O/kcache_after.txt | grep -v "\.tmp" | while read f; do cp -n rainier_amd/kcache/$f $O/kcache_new/ 2>/dev/null; done
ls $O/kcache_new | wc -l
