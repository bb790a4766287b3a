#!/bin/bash
# round 3, GPU call 5: the 4x4x4-MFMA GLM kernel -- parity, speed, counters
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_lowering.py -q -x -k "glm or logistic or cfg4 or 128_columns or mfma" ) > $O/t_glm.log 2>&1; tail -6 $O/t_glm.log
for v in 1 0; do echo "RH_GLM4=$v: $(RH_GLM4=$v timeout 300 python tools/cfg4_probe.py 10000000 256 2 2>&1 | tail -1)"; done
echo "cfg4 as handed over (2e6): $(timeout 300 python tools/cfg4_probe.py 2000000 256 2 reference 2>&1 | tail -1)"
( time timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -q -x -k "cfg4" ) > $O/t_cfg4_sizes.log 2>&1; tail -4 $O/t_cfg4_sizes.log
bash tools/pmc_cfg4.sh > $O/pmc_cfg4.log 2>&1; tail -3 $O/pmc_cfg4.log; cp gpurun_out/pmc_glm.json $O/cfg4_pmc_glm4.json
