#!/bin/bash
# rocprofv3 counter passes over the blur alone (separate --pmc runs: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage: [OUT=dir] tools/pmc_blur.sh <tag> B H W C sigma     -> $OUT/pmc_<tag>_{sq,fetch,write}/ + $OUT/pmc_<tag>.json  (OUT defaults to gpurun_out/)
set -e
export TMPDIR=/tmp
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${OUT:-$root/gpurun_out}
for d in sq fetch write; do if [ -e $out/pmc_${tag}_$d ]; then echo "$out/pmc_${tag}_$d exists: pick a new tag" >&2; exit 2; fi; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $out/pmc_${tag}_sq -- python3 $root/tools/blur_run.py "$@" > $out/pmc_${tag}_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_${tag}_fetch -- python3 $root/tools/blur_run.py "$@" > $out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_${tag}_write -- python3 $root/tools/blur_run.py "$@" > $out/pmc_${tag}_write.log 2>&1
python3 $root/tools/pmc_kernel.py blur $out/pmc_${tag}_sq $out/pmc_${tag}_fetch $out/pmc_${tag}_write > $out/pmc_${tag}.json
cat $out/pmc_${tag}.json
