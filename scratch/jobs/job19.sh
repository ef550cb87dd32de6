cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j19
for r in 1 2; do
  BITS=1 DROP=0.2 python tools/bench_gat.py 2>&1 | tail -1
  SEGGER_EXP_INTERLEAVE=1 BITS=1 DROP=0.2 python tools/bench_gat.py 2>&1 | tail -1 | sed 's/^/interleaved: /'
  SEGGER_AMD_LIB=$PWD/tools/ab_libs/fw3.so BITS=1 DROP=0.2 python tools/bench_gat.py 2>&1 | tail -1
done | tee gpurun_out/j19/gat.txt
for r in 1 2; do
  VARIANTS="default:" ROUNDS=1 STEPS=20 python tools/bench_step.py 2>&1 | grep round
  SEGGER_AMD_LIB=$PWD/tools/ab_libs/fw3.so VARIANTS="fwd_waves3:" ROUNDS=1 STEPS=20 python tools/bench_step.py 2>&1 | grep round
done | tee gpurun_out/j19/step.txt
