#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# sample sclk / power while the cfg 2 bench (or the fp64 ubench) runs
mkdir -p gpurun_out; out=gpurun_out/clock_probe.txt; : > $out
probe() {
  for i in $(seq 1 $1); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' >> $out; echo >> $out
    sleep 0.2
  done
}
echo "## idle" >> $out; probe 2
echo "## bench cfg2 default" >> $out
( RH_FMA_ADDS=0 python bench.py --steps 150 --warmup 2 --no-cpu-baseline > gpurun_out/clock_bench1.json 2>&1 ) & pid=$!
sleep 1.5; probe 6; wait $pid; tail -c 400 gpurun_out/clock_bench1.json >> $out; echo >> $out
echo "## bench cfg2 max-ilp" >> $out
( RH_FMA_ADDS=0 RH_HIPRTC_EXTRA=-mllvm,-amdgpu-sched-strategy=max-ilp python bench.py --steps 150 --warmup 2 --no-cpu-baseline > gpurun_out/clock_bench2.json 2>&1 ) & pid=$!
sleep 1.5; probe 6; wait $pid; tail -c 400 gpurun_out/clock_bench2.json >> $out; echo >> $out
echo "## ubench" >> $out
( for i in 1 2 3 4 5 6; do tools/ubench/fma64_peak > /dev/null; done ) & pid=$!
sleep 0.3; probe 4; wait $pid
cat $out
