#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=..."   ->  tools/ab_libs/NAME.so : the library with the nine gatv2_inst.hip
# objects rebuilt under extra compiler flags (the other objects are taken from build/csrc as they are)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../segger_amd/csrc"
out=../../build/ab/$name; mkdir -p $out ../../tools/ab_libs
for p in 0 1 2; do for d in 0 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 $flags -DSEGGER_INST_PASS=$p -DSEGGER_INST_DTYPE=$d -c gatv2_inst.hip -o $out/inst_p${p}_d${d}.o &
done; done; wait
plain=$(ls ../../build/csrc/*.o | grep -v gatv2_inst)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab_libs/$name.so $plain $out/inst_p*_d*.o
echo built tools/ab_libs/$name.so
