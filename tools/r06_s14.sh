#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/s14; mkdir -p $OUT
timeout -k 10 500 python3 bench.py --gpus 2 --backend gloo --allow-shared-gpu --n-tx 200000 --n-bd 2000 --strong-n-tx 2000000 --strong-n-bd 20000 --steps 3 --warmup 1 --no-c5 > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.log; echo rc=$?
wc -c $OUT/bench_n2_gloo.json; cat $OUT/bench_n2_gloo.json | head -c 1500; echo; grep -i "error\|Traceback" $OUT/bench_n2_gloo.log | head
