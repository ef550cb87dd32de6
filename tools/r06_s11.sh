#!/bin/bash
# the round's committed profiles: kernel traces (prof_round.sh) + the PMC passes of the aggregation kernels
R=$PWD
bash tools/prof_round.sh r06 > gpurun_out/prof_round_r06.log 2>&1 || { tail -30 gpurun_out/prof_round_r06.log; exit 1; }
DROP=0.2 BITS=1 bash tools/pmc.sh r06_pmc tools/bench_gat.py > gpurun_out/r06_pmc.log 2>&1 || { tail -30 gpurun_out/r06_pmc.log; exit 1; }
cd $R
python3 tools/pmc_report.py gpurun_out/r06_pmc > gpurun_out/r06_gatv2_pmc_counters.txt
python3 tools/pmc_to_traffic.py gpurun_out/r06_pmc gpurun_out/hbm_traffic_r06.json > /dev/null
rm -rf gpurun_out/r06_pmc/p*/
head -12 gpurun_out/prof_r06/r06_c2_step_breakdown.txt; head -6 gpurun_out/prof_r06/r06_f32_step_breakdown.txt; head -6 gpurun_out/prof_r06/r06_small_batch_step_graphed.txt; cat gpurun_out/hbm_traffic_r06.json
