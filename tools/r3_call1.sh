#!/bin/bash
# round 3, GPU call 1: the fused launch (correctness + speed), the restored 1e-12 bars, the BASELINE sampler configurations,
# and rocprof evidence for the cfg-5 kernels as they stood at the start of the round.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_a; mkdir -p $O; rm -f gpurun_out/parity_worst.txt gpurun_out/baseline_samplers.txt
HEAD=${1:-unknown}
( time timeout 600 python -m pytest tests/test_gpu_fused.py "tests/test_gpu_parity.py" -q -x -k "fused or tick_engine or cfg2_full or full_driver" ) > $O/t_fused.log 2>&1; tail -5 $O/t_fused.log
( time timeout 300 python bench.py --no-cpu-baseline ) > $O/bench_fused.json 2> $O/bench_fused.err; cut -c1-1500 $O/bench_fused.json; tail -3 $O/bench_fused.err
( RH_FUSE=0 timeout 200 python bench.py --no-cpu-baseline --no-ess --no-inlined ) > $O/bench_unfused.json 2>> $O/bench_fused.err; cut -c1-400 $O/bench_unfused.json
( time timeout 900 python -m pytest tests/test_gpu_reference_lowering.py -q ) > $O/t_reflow.log 2>&1; tail -15 $O/t_reflow.log; cat gpurun_out/parity_worst.txt
( time timeout 1200 python -m pytest tests/test_gpu_baseline_samplers.py -q -s ) > $O/t_samplers.log 2>&1; tail -12 $O/t_samplers.log; cat gpurun_out/baseline_samplers.txt
( time bash tools/profile_side.sh $HEAD r3_cfg5_before cfg5 ) > $O/prof_cfg5.log 2>&1; tail -12 $O/prof_cfg5.log
