#!/bin/bash
# rocprofv3 memory-path counter passes over the band blur (L1 / TLB / L2 write path).  Usage: tools/pmc_band_mem.sh <tag> B H W C sigma
set -e
export TMPDIR=/tmp
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum \
  --kernel-trace --output-format csv -d $out/pmcm_${tag}_a -- python3 $root/tools/blur_run.py "$@" > $out/pmcm_${tag}_a.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_WRITE_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum \
  --kernel-trace --output-format csv -d $out/pmcm_${tag}_b -- python3 $root/tools/blur_run.py "$@" > $out/pmcm_${tag}_b.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum \
  --kernel-trace --output-format csv -d $out/pmcm_${tag}_c -- python3 $root/tools/blur_run.py "$@" > $out/pmcm_${tag}_c.log 2>&1
python3 $root/tools/pmc_kernel.py blur $out/pmcm_${tag}_a $out/pmcm_${tag}_b $out/pmcm_${tag}_c > $out/pmcm_${tag}.json
grep -v dispatches_ $out/pmcm_${tag}.json
