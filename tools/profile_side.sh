#!/bin/bash
# rocprofv3 evidence for the OTHER BASELINE configurations (VERDICT r1 missing #7, weak #8/#10): per workload a
# `--kernel-trace --stats` summary of `bench.py --workload cfgN` and separate `--pmc` passes for its dominant kernel
# (instruction mix | busy / wait cycles | memory-side traffic).  Output: gpurun_out/<name>/<cfg>_{kernel_stats.csv,pmc.json,bench.json}
# usage (via gpurun, from the repo root): bash tools/profile_side.sh <git-head> [name] [workloads...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; HEAD=${1:-unknown}; NAME=${2:-r2_d_side}; shift 2
WL=${@:-"cfg1 cfg3 cfg4 cfg5"}
O=$R/gpurun_out/$NAME; mkdir -p $O
for w in $WL; do
  case $w in
    cfg1) ARGS="--workload cfg1 --steps 200 --warmup 100"; KER=rh_chain_kernel;;
    cfg3) ARGS="--workload cfg3 --steps 100 --warmup 200"; KER=rh_chain_kernel;;
    cfg4) ARGS=""; KER=rh_grad_glm_kernel;;      # full size, static trajectories (a NUTS warm-up at 1e7 rows takes minutes): tools/cfg4_probe.py
    cfg5) ARGS=""; KER=rh_grad_gather_kernel;;   # full size, 32 leapfrog steps of 1024 chains: tools/cfg5_probe.py
  esac
  BENCH="python $R/bench.py $ARGS"
  [ "$w" = "cfg4" ] && BENCH="python $R/tools/cfg4_probe.py 10000000 256 2"
  [ "$w" = "cfg5" ] && BENCH="python $R/tools/cfg5_probe.py 10000 100 1024"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$w -o p -- $BENCH > $O/${w}_bench.json 2> $O/${w}_stats.err
  f=$(find $O/st_$w -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
  rm -rf $O/st_$w
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); mkdir -p $O/${w}_p$i
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/${w}_p$i -o bench -- $BENCH > $O/${w}_p$i/log.txt 2>&1
    f=$(find $O/${w}_p$i -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && mv $f $O/${w}_p$i/bench_counter_collection.csv
    find $O/${w}_p$i -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
  done
  python $R/profiles/summarize.py $KER 64 $O/${w}_pmc.json $O/${w}_p1 $O/${w}_p2 $O/${w}_p3 $O/${w}_p4 --meta git_head=$HEAD workload=$w > /dev/null 2>&1
  [ "$w" = "cfg5" ] && python $R/profiles/summarize.py rh_tick_kernel 64 $O/${w}_pmc_tick.json $O/${w}_p1 $O/${w}_p2 $O/${w}_p3 --meta git_head=$HEAD workload=$w > /dev/null 2>&1
  rm -rf $O/${w}_p?
  echo "== $w"; tail -2 $O/${w}_bench.json | cut -c1-600; head -4 $O/${w}_kernel_stats.csv
  python - <<PY
import json
d=json.load(open("$O/${w}_pmc.json"))
print({k:(round(v["mean_per_launch"]),round(v["mean_duration_us"],1)) for k,v in d["counters"].items()})
PY
done
