#!/bin/bash
# A/B of two builds of the library inside one GPU call: tools/ab_lib.sh tools/_build/libbgan_old.so "G4,D3" [arch] [batch]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/exp
for rep in 1 2; do
  BGAN_HIP_LIB=$PWD/$1 python tools/bench_conv.py --arch ${3:-celeba64} --batch ${4:-256} --iters 20 --only "$2" > gpurun_out/exp/ab_old$rep.log 2>&1
  python tools/bench_conv.py --arch ${3:-celeba64} --batch ${4:-256} --iters 20 --only "$2" > gpurun_out/exp/ab_new$rep.log 2>&1
done
for f in old1 new1 old2 new2; do echo "== $f"; grep -E "fwd|dgrad|wgrad" gpurun_out/exp/ab_$f.log | cut -c1-110; done
