#!/bin/bash
# usage: pmc.sh <outdir> <script> ; collects PMC counters in separate passes (kernel-trace only)
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 $R/$2 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc FETCH_SIZE TCC_HIT_sum -- python3 $R/$2 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -- python3 $R/$2 > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p4 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -- python3 $R/$2 > $OUT/p4.log 2>&1
ls $OUT/*/*/ | head -30
