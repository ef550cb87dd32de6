#!/bin/bash
# round 6, GPU session 2: the rewritten GELU and the window-64 by-destination order: kernels alone, counters, whole step A/B
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT/valu_budget
MODE=time python3 tools/valu_budget.py > $OUT/valu_budget/time.log 2>&1 || { tail -20 $OUT/valu_budget/time.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
MODE=count rocprofv3 --kernel-trace --output-format csv -d $OUT/valu_budget/pmc1 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -- python3 $R/tools/valu_budget.py > $OUT/valu_budget/count.log 2>&1 || { tail -20 $OUT/valu_budget/count.log; exit 1; }
cd $R
python3 tools/valu_budget_fit.py > $OUT/valu_budget/fit.txt 2>&1; cat $OUT/valu_budget/fit.txt
VARIANTS="natural:ROW_ORDER=0;win64:ROW_ORDER=64" ROUNDS=3 python3 tools/bench_step.py > $OUT/step_ab_order.log 2>&1; cat $OUT/step_ab_order.log
python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_s2.log 2>&1; tail -5 $OUT/pytest_gpu_s2.log
