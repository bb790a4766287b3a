#!/bin/bash
# rocprofv3 evidence for the bench command (cfg 2, dominant kernel rh_grad_fused_kernel) at the current HEAD:
#   1. `--kernel-trace --stats` summary of `python bench.py --steps 20 --warmup 5`  -> kernel_stats.csv
#   2. separate `--pmc` passes (FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ busy/wait | TCC hit/miss), as the
#      guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with runtime traces)
#   3. condensed into pmc_grad_kernel.json, stamped with the git head and the sha of the generated kernel source so that
#      bench.py only quotes `roofline.traffic` from a profile of byte-identical code.
# usage (from the repo root, via gpurun): bash tools/pmc_cfg2.sh <git-head> [outdir-name]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; HEAD=${1:-unknown}; NAME=${2:-r4_cfg2}
O=$R/gpurun_out/$NAME; mkdir -p $O
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ess --no-inlined --no-configs --no-live-traffic"
$BENCH > $O/bench_noprof.json 2> $O/bench_noprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/bench_stats_run.json 2> $O/stats.err
f=$(find $O/stats -name "bench_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); mkdir -p $O/p$i
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/p$i -o bench -- $BENCH > $O/p$i/log.txt 2>&1
  f=$(find $O/p$i -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/p$i/bench_counter_collection.csv
done
python - "$O/p1/bench_counter_collection.csv" > $O/p1_kernels.txt 2>&1 <<'PY'
import csv, sys, collections
c = collections.Counter(r["Kernel_Name"] for r in csv.DictReader(open(sys.argv[1])))
for k, n in c.most_common(): print(n, k)
PY
tail -3 $O/p1/log.txt > $O/p1_log_tail.txt 2>/dev/null
SHA=$(python -c "import json;print(json.load(open('$O/bench_noprof.json'))['config']['generated_source_sha16'])")
python $R/profiles/summarize.py rh_grad_fused_kernel 256 $O/pmc_grad_kernel.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 \
  --meta git_head=$HEAD generated_source_sha16=$SHA rows=1000000 chains_per_gpu=1024 > $O/summary.txt 2>&1
python $R/profiles/summarize.py rh_tick_kernel 16 $O/pmc_tick_kernel.json $O/p3 $O/p4 --meta git_head=$HEAD >> $O/summary.txt 2>&1
rm -rf $O/stats $O/p?/[!bl]* $O/p? 2>/dev/null
tail -30 $O/summary.txt | head -5; cat $O/kernel_stats.csv | head -6; cut -c1-300 $O/bench_noprof.json
