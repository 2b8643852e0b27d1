#!/bin/bash
# Regenerates the training-step evidence of a round on the GPU box (run from the repository root through gpurun):
#   tools/evidence_steps.sh <rNN_stage>
# -> gpurun_out/<tag>/evidence_steps/: the default bench lines of C2 / C4 / C1 (with roofline and cpu_baseline), a rocprofv3
# --stats run of the C2 and C4 commands with their summaries (ready to copy into profiles/ as <tag>_*), and smoke().
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
. tools/_fresh.sh "$@"
timeout -k 10 300 python bench.py 2>$OUT/bench_c2.err | tail -1 > $OUT/${tag}_bench.json; cut -c1-200 $OUT/${tag}_bench.json
timeout -k 10 300 python bench.py --arch celeba128 2>$OUT/bench_c4.err | tail -1 > $OUT/${tag}_bench_c4_celeba128.json; cut -c1-200 $OUT/${tag}_bench_c4_celeba128.json
timeout -k 10 300 python bench.py --arch mnist 2>$OUT/bench_c1.err | tail -1 > $OUT/${tag}_bench_c1_mnist.json; cut -c1-200 $OUT/${tag}_bench_c1_mnist.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${tag}_bench_profiled_run.json 2> $OUT/prof_c2.log)
python3 tools/rocprof_summary.py $OUT/prof_c2 $OUT/${tag}_kernel_stats.md "$tag: rocprofv3 kernel stats of bench.py --steps 10 --warmup 3 (C2: celeba64, batch 256)"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -- python3 $root/bench.py --arch celeba128 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${tag}_bench_c4_profiled_run.json 2> $OUT/prof_c4.log)
python3 tools/rocprof_summary.py $OUT/prof_c4 $OUT/${tag}_kernel_stats_c4.md "$tag: rocprofv3 kernel stats of bench.py --arch celeba128 --steps 10 --warmup 3 (C4: batch 128)"
# N = 1 of the scaling run with every collective of the N-rank step issued through RCCL (one rank): comparable with the plain C2 line above
BGAN_DP_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline 2>$OUT/bench_dp1.err | tail -1 > $OUT/${tag}_bench_dp1_torch.json; cut -c1-200 $OUT/${tag}_bench_dp1_torch.json
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
