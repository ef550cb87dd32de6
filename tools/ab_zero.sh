#!/bin/bash
# A/B of the zero fill riding on the tx-neighbors-tx source pass (vs the separate zero_rows kernel), same box
for on in True False True False; do
  python -c "
import sys, runpy
import segger_amd.ops as ops
if not $on:
    _orig = ops.gatv2_bwd_launch
    def patched(*a, zero_rows_out=None, grad_xl_zeroed=False, **k):
        r = _orig(*a, **k); patched.zero_filled = False; return r
    patched.zero_filled = False
    ops.gatv2_bwd_launch = patched
sys.argv = ['bench.py', '--no-strong', '--no-f32', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused zero fill $on', round(d['ms_per_step'],3))"
done
