#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/s5; mkdir -p $OUT
timeout -k 10 300 python3 tools/overlap_probe.py > $OUT/overlap.log 2>&1; cat $OUT/overlap.log | grep -v Warning
