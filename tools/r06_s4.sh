#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/s4; mkdir -p $OUT
VARIANTS="natural:ROW_ORDER=0;dst64:ROW_ORDER=64;both64:ROW_ORDER=64,ORDER_FWD=1" ROUNDS=4 STEPS=20 python3 tools/bench_step.py > $OUT/step_ab.log 2>&1; grep round $OUT/step_ab.log
