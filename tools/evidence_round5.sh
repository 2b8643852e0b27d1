#!/bin/bash
# Round-5 evidence on the GPU box, everything into a fresh gpurun_out/<tag>/evidence_round5/ (run through gpurun, ~6 minutes):
#   tools/evidence_round5.sh <r04_stage>
# PMC traffic of the C2 and C4 steps (+ the per-launch filter-gradient table), the matrix-pipe counters of both, the step bench
# lines (after traffic_update, so that roofline.traffic is filled in), kernel-trace timelines of the MNIST step (replay / eager),
# host time per step (eager / replay / graph), the one-rank collective rehearsals.
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
. tools/_fresh.sh "$@"
export OUT
for c in "c2 celeba64 256" "c4 celeba128 128" "c1 mnist 64"; do set -- $c
  tools/pmc_step.sh $1 --arch $2 > $OUT/pmc_$1.log 2>&1 || { tail -5 $OUT/pmc_$1.log; exit 1; }
  python3 tools/traffic_update.py $2 $3 $OUT/pmc_$1_traffic.json
  cp $OUT/pmc_$1.md $OUT/${tag}_pmc_traffic_$1.md; cp $OUT/pmc_$1_traffic.json.wgrad.md $OUT/${tag}_pmc_wgrad_launches_$1.md
  tools/pmc_mfma.sh $1 bench.py --arch $2 --steps 5 --warmup 3 --no-cpu-baseline --no-profile > $OUT/mfma_$1.log 2>&1
  cp $OUT/pmc_mfma_$1.md $OUT/${tag}_pmc_mfma_$1.md
  rm -rf $OUT/pmc_$1_stats $OUT/pmc_$1_fetch $OUT/pmc_$1_write $OUT/pmc_mfma_$1
done
cp profiles/hbm_traffic.json $OUT/${tag}_hbm_traffic.json
for a in mnist celeba64; do python3 tools/host_time.py $a 2>/dev/null | tail -1 > $OUT/${tag}_host_time_$a.json; cut -c1-300 $OUT/${tag}_host_time_$a.json; done
export TMPDIR=/tmp
for m in replay eager; do
  (cd /tmp && BGAN_NO_STEP_REPLAY=$([ $m = eager ] && echo 1 || echo 0) rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_mnist_$m -- python3 $root/bench.py --arch mnist --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/trace_mnist_$m.json 2> $OUT/trace_mnist_$m.log)
  python3 tools/step_timeline.py $OUT/trace_mnist_$m $OUT/${tag}_timeline_mnist_$m.md 0 "$tag: GPU timeline of bench.py --arch mnist ($m)" > /dev/null
  rm -rf $OUT/trace_mnist_$m
done
for route in torch abi; do
  BGAN_DP_COLLECTIVE=$route BGAN_DP_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu-baseline 2>$OUT/dp1_$route.err | tail -1 > $OUT/${tag}_bench_dp1_$route.json; cut -c1-120 $OUT/${tag}_bench_dp1_$route.json
done
echo done
