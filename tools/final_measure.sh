#!/bin/bash
# consolidated end-of-round measurements (all five BASELINE configs), one JSON line each -> gpurun_out/final/
mkdir -p gpurun_out/final; O=gpurun_out/final
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --strict --no-cpu-baseline > $O/bench_cfg2_strict.json 2>> $O/bench_cfg2.err
python bench.py --workload cfg1 --no-cpu-baseline --steps 400 --warmup 200 > $O/cfg1_1024.json 2>&1
python bench.py --workload cfg1 --no-cpu-baseline --steps 400 --warmup 200 --chains-per-gpu 32768 > $O/cfg1_32768.json 2>&1
python bench.py --workload cfg3 --no-cpu-baseline --steps 400 --warmup 200 > $O/cfg3_1024.json 2>&1
python bench.py --workload cfg3 --no-cpu-baseline --steps 400 --warmup 200 --chains-per-gpu 32768 > $O/cfg3_32768.json 2>&1
python bench.py --workload cfg3 --sampler nuts --no-cpu-baseline --steps 200 --warmup 200 > $O/cfg3_nuts_1024.json 2>&1
timeout 400 python tools/cfg4_probe.py 10000000 256 2 > $O/cfg4_full.txt 2>&1
timeout 400 python tools/cfg5_probe.py 10000 100 1024 > $O/cfg5_full.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tail -n 2 $O/*.json $O/*.txt | cut -c1-600
