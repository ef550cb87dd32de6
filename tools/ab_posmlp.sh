#!/bin/bash
# A/B of the fused positional embedder on the C2 bench, same box: tools/ab_posmlp.sh
for fused in True False True False; do
  python -c "
import sys, runpy
import segger_amd.ist_encoder as m
m.FUSED_POSMLP = $fused
sys.argv = ['bench.py', '--no-strong', '--no-f32', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused $fused', round(d['ms_per_step'],3), round(d['predict']['ms_per_batch'],3))"
done
