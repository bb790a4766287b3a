#!/bin/bash
# GPU call H of round 5: the tests added after the final tier (glm4r in the suite, the fused big-mode tick under a diagonal mass) and
# cfg 5 at the 64-split default.  -> gpurun_out/r5_h/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5_h; mkdir -p $O
ls rainier_amd/kcache | sort > $O/kcache_before.txt
( time timeout -s INT 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "glm4r or big_mode_chain or gather_mode or cfg5" ) > $O/tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/tests.log | cut -c1-300 | tail -8
( RH_PROBE_SPLITS=0,48 timeout 300 python tools/cfg5_probe.py 10000 100 1024 0 ) > $O/cfg5_probe.txt 2>&1
grep '^{"G"' $O/cfg5_probe.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   splits %2d: gather %.3f ms, per step %.3f ms' % (d['splits'], d['grad_kernel_ms'], d['all_ms']))"
mkdir -p $O/kcache_new; ls rainier_amd/kcache | sort > $O/kcache_after.txt
comm -13 $O/kcache_before.txt $O/kcache_after.txt | grep -v "\.tmp" | while read f; do cp -n rainier_amd/kcache/$f $O/kcache_new/ 2>/dev/null; done
ls $O/kcache_new | wc -l
