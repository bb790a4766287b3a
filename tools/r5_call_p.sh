#!/bin/bash
# GPU call P of round 5 (the budget's last five minutes): a sixth fuzz sweep with the sampler runs, fresh seeds, in two halves so that
# a half that finishes leaves its summary line whatever happens to the other.  -> gpurun_out/r5_p/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_p; mkdir -p $O
T0=$SECONDS
( SWEEP_CHAINS=1 timeout -s INT 130 python tools/gpu_fuzz_sweep.py run 456 468 ) > $O/fuzz_sweep_chains_456_468.txt 2>&1
tail -2 $O/fuzz_sweep_chains_456_468.txt | cut -c1-300
LEFT=$((255 - (SECONDS - T0)))
if [ $LEFT -gt 40 ]; then
  ( SWEEP_CHAINS=1 timeout -s INT $LEFT python tools/gpu_fuzz_sweep.py run 468 480 ) > $O/fuzz_sweep_chains_468_480.txt 2>&1
  tail -2 $O/fuzz_sweep_chains_468_480.txt | cut -c1-300
fi
echo "elapsed $((SECONDS - T0)) s"
