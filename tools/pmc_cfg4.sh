#!/bin/bash
export RH_DIAG=1   # the engine reads its experiment switches only in a process that asks for them (csrc/rir.hpp: rh::knob)
# PMC passes over the cfg-4-shaped probe (rh_grad_glm_kernel); results condensed into gpurun_out/pmc_glm.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_cfg4; mkdir -p $O
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1)); mkdir -p $O/p$i
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/p$i -o bench -- python $R/tools/cfg4_probe.py 2000000 256 2 > $O/p$i/log.txt 2>&1
  f=$(find $O/p$i -name "bench_counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/p$i/bench_counter_collection.csv
done
python $R/profiles/summarize.py ${RH_PMC_KERNEL:-rh_grad_glm4_kernel} 8 $R/gpurun_out/pmc_glm.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 2>&1 | tail -80
