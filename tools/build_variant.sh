#!/bin/bash
# Builds a variant of the library with extra -D flags for ONE source file, linked against the other in-tree objects:
#   tools/build_variant.sh <name> <source.hip> "<-Dflags>"   ->  tools/_build/libbgan_<name>.so   (select with BGAN_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
mkdir -p tools/_build
obj=tools/_build/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off $flags -c blurred-gan_amd/csrc/$src -o $obj
others=$(ls blurred-gan_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj $others -ldl -o tools/_build/libbgan_$name.so
echo tools/_build/libbgan_$name.so
