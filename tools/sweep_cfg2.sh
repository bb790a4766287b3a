#!/bin/bash
# cfg 2 tuning sweep (one line each): "<env> | <bench flags>"
mkdir -p gpurun_out
out=gpurun_out/sweep_cfg2.txt; : > $out
while IFS='|' read -r envs flags; do
  [ -z "$flags" ] && continue
  echo "## $envs | $flags" >> $out
  env $envs timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline $flags 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print(d['ms_per_step'], d['value'])
except Exception as e: print('ERR', l[-300:])" >> $out
done < "${1:-tools/sweep_cfg2.list}"
cat $out
