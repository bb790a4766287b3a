#!/bin/bash
# round 3, GPU call 4: cfg 2 workgroup-shape sweep, cfg 4 counters + fp64 MFMA ubench counters, cfg 1 at 1024 chains, centred cfg 5, multi-GPU readiness tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_d; mkdir -p $O; rm -f gpurun_out/baseline_samplers.txt
export RH_FUSE=0
for u in 4 2; do for gs in 24 32 48 64; do
  ( timeout 200 python bench.py --no-cpu-baseline --no-ess --no-inlined --steps 40 --warmup 20 --grad-unroll $u --grad-splits $gs ) > $O/b_${u}_${gs}.json 2> $O/b_${u}_${gs}.err
  python -c "
import json
d=json.loads(open('$O/b_${u}_${gs}.json').read().strip().splitlines()[-1]); r=d['roofline']
print('unroll $u splits $gs: ms_per_step %.3f avg_launch_ms %.4f frac %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac']))"
done; done
unset RH_FUSE
( time timeout 600 python -m pytest tests/test_gpu_baseline_samplers.py -q -s -k "centred" ) > $O/t_centred.log 2>&1; tail -4 $O/t_centred.log; cat gpurun_out/baseline_samplers.txt
( time timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_jni_shim.py -q ) > $O/t_multi.log 2>&1; tail -6 $O/t_multi.log
# cfg 1 at the BASELINE chain count: packed (4 chains per wavefront) vs one chain per wavefront, 1 / 2 waves per SIMD requested
for v in "" "RH_PACK=0" "RH_PACK=0 RH_CHAIN_WAVES=1" "RH_CHAIN_WAVES=1"; do
  echo "cfg1 [$v]: $(env $v timeout 120 python bench.py --workload cfg1 --steps 400 --warmup 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e steps/s' % d['value'])")"
done
bash tools/pmc_cfg4.sh > $O/pmc_cfg4.log 2>&1; tail -3 $O/pmc_cfg4.log; cp gpurun_out/pmc_glm.json $O/cfg4_pmc_glm.json
# fp64 matrix-pipe ubench with counters: issue interval of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64
cd /tmp && export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/ubench/fma64_cycles mfma > $GRAFT_REPO_ROOT/$O/mfma_ubench.txt 2>&1; cat $GRAFT_REPO_ROOT/$O/mfma_ubench.txt
for ctrs in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  d=$GRAFT_REPO_ROOT/$O/mfma_pmc_$(echo $ctrs | cut -c1-12 | tr ' ' _); mkdir -p $d
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o ub -- $GRAFT_REPO_ROOT/tools/ubench/fma64_cycles mfma > $d/log.txt 2>&1
  f=$(find $d -name "ub_counter_collection.csv" | head -1); [ -n "$f" ] && python - <<PY
import csv, collections
acc=collections.OrderedDict()
for r in csv.DictReader(open("$f")):
    k=(r["Dispatch_Id"], r["Kernel_Name"][:40]); acc.setdefault(k, {})[r["Counter_Name"]]=float(r["Counter_Value"]); acc[k]["dur_us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for k,v in acc.items(): print(k, v)
PY
done
