#!/bin/bash
# usage: pmc_quick.sh <outdir> <script> : one SQ pass (instruction counts / busy / wait cycles per kernel), kernel-trace only
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 $R/$2 > $OUT/p1.log 2>&1
cd $R; python tools/pmc_report.py gpurun_out/$1 | grep -A9 "gatv2_fwd\|gatv2_bwd" 
