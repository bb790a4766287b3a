#!/bin/bash
# GPU call F of round 5: the wide fuzz sweep on fresh seeds (every case, both math modes, both engines, against the oracle at
# 1e-12 * sum|term|) and rh_grad_glm4r_kernel with one chain group per wavefront (three wavefronts per SIMD).  -> gpurun_out/r5_f/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_f; mkdir -p $O
for v in "RH_GLM4R=1" "RH_GLM4R=1 RH_GLM4R_JG=1" "RH_GLM4R=1 RH_GLM4R_JG=1 RH_GLM4R_W=8"; do
  ( env $v timeout 200 python tools/cfg4_probe.py 2000000 256 2 ) > "$O/cfg4_probe_$(echo $v | tr ' =' '__').txt" 2>&1; echo "-- $v"; tail -1 "$O/cfg4_probe_$(echo $v | tr ' =' '__').txt" | cut -c1-260
done
( time timeout 900 python tools/gpu_fuzz_sweep.py run 300 332 ) > $O/fuzz_sweep_300_332.txt 2>&1
grep -E "FAIL|ERROR|sweep" $O/fuzz_sweep_300_332.txt | cut -c1-300 | tail -12
