#!/bin/bash
# A/B of the split first-layer projection (per-gene table + positional GEMM) on the C2 bench, same box
for split in True False True False; do
  python -c "
import sys, runpy, torch
import segger_amd.ist_encoder as m
_init = m.ISTEncoder.__init__
def init(self, *a, **k):
    _init(self, *a, **k); self.split_first_layer = $split
m.ISTEncoder.__init__ = init
sys.argv = ['bench.py', '--no-strong', '--no-f32', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $split', round(d['ms_per_step'],3), round(d['predict']['ms_per_batch'],3))"
done
