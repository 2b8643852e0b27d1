#!/bin/bash
# A/B of an environment knob inside one process-per-setting: tools/exp_env.sh VAR "v1 v2 ..." "LAYERS" [arch] [batch]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/exp
V=$1
python tools/bench_conv.py --arch ${4:-celeba64} --batch ${5:-256} --iters 20 --only "$3" > gpurun_out/exp/env_base.log 2>&1
echo "== base"; grep -E "fwd|dgrad|wgrad" gpurun_out/exp/env_base.log | cut -c1-100
for x in $2; do
  env $V=$x python tools/bench_conv.py --arch ${4:-celeba64} --batch ${5:-256} --iters 20 --only "$3" > gpurun_out/exp/env_$x.log 2>&1
  echo "== $V=$x"; grep -E "fwd|dgrad|wgrad" gpurun_out/exp/env_$x.log | cut -c1-100
done
