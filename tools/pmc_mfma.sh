#!/bin/bash
# Matrix-pipe utilisation per kernel symbol from a hardware counter pass (north_star: "MFMA utilisation against peak"):
#   tools/pmc_mfma.sh <tag> <program args ...>      e.g.  tools/pmc_mfma.sh r04_c2 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-profile
# One rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE: SQ and GRBM slots, no TCC counter, no
# trace domain besides --kernel-trace); the program follows `--` directly (python3 <script>), as the pool requires.
# -> $OUT/pmc_mfma_<tag>/ (raw CSVs), $OUT/pmc_mfma_<tag>.md (per-kernel table)
set -e
export TMPDIR=/tmp
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${OUT:-$root/gpurun_out}
if [ -e $out/pmc_mfma_$tag ]; then echo "$out/pmc_mfma_$tag exists: pick a new tag" >&2; exit 2; fi
prog=$1; shift
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma_$tag -- python3 $root/$prog "$@" > $out/pmc_mfma_${tag}_stdout.txt 2> $out/pmc_mfma_$tag.log)
python3 $root/tools/pmc_mfma_report.py $out/pmc_mfma_$tag $out/pmc_mfma_$tag.md "$tag: $prog $*"
