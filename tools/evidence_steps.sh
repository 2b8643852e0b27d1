#!/bin/bash
# Regenerates the training-step evidence of a round on the GPU box (run from the repository root through gpurun): the default
# bench lines of C2 / C4 / C1 (with roofline and cpu_baseline), a rocprofv3 --stats run of the C2 command, and smoke().
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
root=$(pwd)
export TMPDIR=/tmp
timeout -k 10 300 python bench.py 2>gpurun_out/bench_c2.err | tail -1 > gpurun_out/bench_c2.json; cut -c1-200 gpurun_out/bench_c2.json
timeout -k 10 300 python bench.py --arch celeba128 2>gpurun_out/bench_c4.err | tail -1 > gpurun_out/bench_c4.json; cut -c1-200 gpurun_out/bench_c4.json
timeout -k 10 300 python bench.py --arch mnist 2>gpurun_out/bench_c1.err | tail -1 > gpurun_out/bench_c1.json; cut -c1-200 gpurun_out/bench_c1.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_c2 -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $root/gpurun_out/prof_c2.log 2>&1)
tail -1 gpurun_out/prof_c2.log | cut -c1-160
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
