#!/bin/bash
# Regenerates the blur evidence of a round on the GPU box (run from the repository root through gpurun):
#   tools/evidence_blur.sh <rNN_stage>
# -> gpurun_out/<tag>/evidence_blur/: parity tests, rocprofv3 counter passes at 31 / 143 / 255 taps, the refreshed blur256 entry
# of profiles/hbm_traffic.json (copied there as <tag>_hbm_traffic.json), the three `bench.py --arch blur256` lines (roofline +
# cpu_baseline) and a rocprofv3 --stats run of the 143-tap line with its summary.
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
. tools/_fresh.sh "$@"
timeout -k 10 300 python -m pytest tests/test_blur_gpu.py -x -q 2>&1 | tail -2
for s in "31 5" "143 23.5" "255 42.34"; do set -- $s; echo "== pmc $1 taps"; OUT=$OUT timeout -k 10 300 tools/pmc_blur.sh f$1 64 256 256 3 $2 > $OUT/pmc_f$1.out 2>&1; tail -3 $OUT/pmc_f$1.out | head -2; cp $OUT/pmc_f$1.json $OUT/${tag}_blur_$1_pmc.json; done
python3 tools/pmc_blur_update.py 31=$OUT/pmc_f31.json 143=$OUT/pmc_f143.json 255=$OUT/pmc_f255.json > $OUT/pmc_blur_update.out
cp profiles/hbm_traffic.json $OUT/${tag}_hbm_traffic.json
for s in 5 23.5 42.34; do echo "== bench sigma $s"; timeout -k 10 300 python bench.py --arch blur256 --sigma $s 2>/dev/null | tail -1 > $OUT/${tag}_bench_blur256_sigma$s.json; cut -c1-240 $OUT/${tag}_bench_blur256_sigma$s.json; done
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_blur143 -- python3 $root/bench.py --arch blur256 --sigma 23.5 --no-cpu-baseline > $OUT/prof_blur143.log 2>&1)
python3 tools/rocprof_summary.py $OUT/prof_blur143 $OUT/${tag}_kernel_stats_blur256_sigma23.5.md "$tag: rocprofv3 kernel stats of bench.py --arch blur256 --sigma 23.5"
tail -1 $OUT/prof_blur143.log | cut -c1-200
