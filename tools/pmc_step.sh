#!/bin/bash
# rocprofv3 passes over bench.py itself (separate --pmc runs: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains
# besides --kernel-trace):  tools/pmc_step.sh <tag> [bench.py args]
#   -> $OUT/pmc_<tag>_{stats,fetch,write}/, $OUT/pmc_<tag>_kernels.json (bench's own per-tag table), $OUT/pmc_<tag>.json,
#      $OUT/pmc_<tag>_traffic.json (conv launches matched with bench's tag sequence; feed it to tools/traffic_update.py)
# OUT defaults to gpurun_out/; the three profile directories must not exist yet (a stale CSV must never be summarised).
set -e
export TMPDIR=/tmp
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${OUT:-$root/gpurun_out}
for d in stats fetch write; do if [ -e $out/pmc_${tag}_$d ]; then echo "$out/pmc_${tag}_$d exists: pick a new tag" >&2; exit 2; fi; done
common="--steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/pmc_${tag}_stats -- python3 $root/bench.py $common --kernels-out $out/pmc_${tag}_kernels.json "$@" > $out/pmc_${tag}_bench.json 2> $out/pmc_${tag}_stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_${tag}_fetch -- python3 $root/bench.py $common "$@" > /dev/null 2> $out/pmc_${tag}_fetch.log
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_${tag}_write -- python3 $root/bench.py $common "$@" > /dev/null 2> $out/pmc_${tag}_write.log
PMC_SKIP=0 python3 $root/tools/pmc_kernel.py "" $out/pmc_${tag}_fetch $out/pmc_${tag}_write > $out/pmc_${tag}.json
python3 $root/tools/pmc_report.py $out/pmc_${tag}.json $out/pmc_${tag}_stats $out/pmc_${tag}_kernels.json > $out/pmc_${tag}.md
python3 $root/tools/pmc_traffic.py $out/pmc_${tag}_fetch $out/pmc_${tag}_write $out/pmc_${tag}_kernels.json $out/pmc_${tag}_traffic.json > /dev/null
head -40 $out/pmc_${tag}.md
