#!/bin/bash
# Builds experiment variants of one kernel file: tools/exp_build.sh conv_wgrad "1 2 4 8 15" -> tools/_build/libbgan_exp<N>.so
# (compiled with -DBG_EXP=<N>; load with BGAN_HIP_LIB=...).  Timing experiments only: knocked-out variants compute garbage.
set -e
cd "$(dirname "$0")/.."
F=$1; shift
mkdir -p tools/_build
for n in $1; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DBG_EXP=$n -c blurred-gan_amd/csrc/$F.hip -o tools/_build/${F}_exp$n.o 2>/dev/null
    objs=""
    for o in runtime blur conv_igemm conv_rows conv_wgrad misc; do
      if [ $o = $F ]; then objs="$objs tools/_build/${F}_exp$n.o"; else objs="$objs blurred-gan_amd/csrc/$o.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/_build/libbgan_exp$n.so ) &
done
wait
ls tools/_build
