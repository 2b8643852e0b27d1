#!/bin/bash
# Regenerates the blur evidence of a round on the GPU box (run from the repository root through gpurun):
# parity tests, rocprofv3 counter passes at 31 / 143 / 255 taps, the refreshed blur256 entry of profiles/r02_hbm_traffic.json
# (copied to gpurun_out/ as well), the three `bench.py --arch blur256` lines and a rocprofv3 --stats run of the 143-tap line.
# Copy what it leaves under gpurun_out/ into profiles/ afterwards (see profiles/r02_e_*).
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
root=$(pwd)
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_blur_gpu.py -x -q 2>&1 | tail -2
for s in "f31 5" "f143 23.5" "f255 42.34"; do set -- $s; echo "== pmc $1"; timeout -k 10 300 tools/pmc_blur.sh $1 64 256 256 3 $2 > gpurun_out/pmc_$1.out 2>&1; tail -3 gpurun_out/pmc_$1.out | head -2; done
python3 tools/pmc_blur_update.py 31=gpurun_out/pmc_f31.json 143=gpurun_out/pmc_f143.json 255=gpurun_out/pmc_f255.json > gpurun_out/pmc_blur_update.out
cp profiles/r02_hbm_traffic.json gpurun_out/r02_hbm_traffic.json
for s in 5 23.5 42.34; do echo "== bench sigma $s"; timeout -k 10 200 python bench.py --arch blur256 --sigma $s 2>/dev/null | tail -1 > gpurun_out/bench_blur256_sigma$s.json; cut -c1-240 gpurun_out/bench_blur256_sigma$s.json; done
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_blur143 -- python3 $root/bench.py --arch blur256 --sigma 23.5 > $root/gpurun_out/prof_blur143.log 2>&1
tail -1 $root/gpurun_out/prof_blur143.log | cut -c1-200
