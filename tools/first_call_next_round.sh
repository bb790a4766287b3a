#!/bin/bash
# A first GPU call for the next round (≈ 12 GPU-minutes): where round 3 stopped, re-measured, plus the two reproducers of the
# spill fault.  usage, from the repo root:  gpurun --timeout 1500 -- "bash tools/first_call_next_round.sh $(git rev-parse --short HEAD)"
cd $GRAFT_REPO_ROOT; O=gpurun_out/next_a; mkdir -p $O; rm -f gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt
HEAD=${1:-unknown}
# 1. the whole GPU tier (incl. the fuzz through the kernels)
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/t_all.log 2>&1; tail -4 $O/t_all.log
# 2. the wide fuzz sweep on both engines (needs `python tools/gpu_fuzz_sweep.py prebuild 100 220` run locally first: ~6 min of CPU)
( time timeout 200 python tools/gpu_fuzz_sweep.py run 100 220 ) > $O/sweep.txt 2>&1; tail -2 $O/sweep.txt
# 3. the reproducers of the spill fault, guard off (each in its own process: the second one may take a GPU memory fault)
( RH_KEEP_UNROLL=1 GU=8 GK=8 NP=3 timeout 120 python tools/fuzz_diag.py 1 4096 ) > $O/repro_grad_kernel.txt 2>&1; grep "engine 2 splits 1" -A1 $O/repro_grad_kernel.txt | tail -2
( RH_KEEP_UNROLL=1 timeout 120 python tools/fuzz_case_chain.py 120 4096 4 ) > $O/repro_density_kernel.txt 2>&1; tail -2 $O/repro_density_kernel.txt
# 4. the bench command: kernel stats + PMC passes, then the default line
bash tools/pmc_cfg2.sh $HEAD next_cfg2 > $O/pmc_cfg2.log 2>&1; tail -6 $O/pmc_cfg2.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
cp gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt $O/ 2>/dev/null
