#!/bin/bash
# usage: pmc_sq.sh <outdir> <script> : two SQ counter passes (kernel-trace only) + report
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 $R/$2 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES -- python3 $R/$2 > $OUT/p2.log 2>&1
cd $R; python tools/pmc_report.py gpurun_out/$1
