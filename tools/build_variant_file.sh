#!/bin/bash
# tools/build_variant_file.sh NAME FILE.hip "-DFLAG=..."  ->  tools/ab_libs/NAME.so : the library with ONE translation unit
# rebuilt under extra compiler flags (the other objects are taken from build/csrc as they are)
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../segger_amd/csrc"
out=../../build/ab/$name; mkdir -p $out ../../tools/ab_libs
base=${file%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 $flags -c $file -o $out/$base.o
rest=$(ls ../../build/csrc/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab_libs/$name.so $rest $out/$base.o
echo built tools/ab_libs/$name.so
