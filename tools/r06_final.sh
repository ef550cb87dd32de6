#!/bin/bash
# the round's closing run: GPU suite, smoke, the driver-shaped bench command
R=$PWD; OUT=$R/gpurun_out/final; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
( time python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log ) 2> $OUT/bench.time || { tail -30 $OUT/bench.log; exit 1; }
wc -c $OUT/bench.json; cat $OUT/bench.time; cp bench_details.json $OUT/bench_details.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['dominant']['ms_per_launch'], d['roofline']['dominant']['frac'], d['f32']['ms_per_step'], d['strong']['graphed']['ms_per_step'], d['strong']['graphed_f32']['ms_per_step'], d['auroc']['met'])
PY
