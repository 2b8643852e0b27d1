#!/bin/bash
# runs bench_conv for the stock library and each experiment variant: tools/exp_run.sh "1 2 4" "G4,D3,G1" [op-filter]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/exp
python tools/bench_conv.py --arch celeba64 --batch 256 --iters 20 --only "$2" > gpurun_out/exp/base.log 2>&1
for n in $1; do
  BGAN_HIP_LIB=$PWD/tools/_build/libbgan_exp$n.so python tools/bench_conv.py --arch celeba64 --batch 256 --iters 20 --only "$2" > gpurun_out/exp/exp$n.log 2>&1
done
for f in base $(for n in $1; do echo exp$n; done); do echo "== $f"; grep -E "${3:-fwd|dgrad|wgrad}" gpurun_out/exp/$f.log | grep -v TOTAL | cut -c1-60; done
