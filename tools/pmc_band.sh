#!/bin/bash
# rocprofv3 SQ counter passes over the band blur (wait / issue / FIFO-full breakdown).  Usage: tools/pmc_band.sh <tag> B H W C sigma
set -e
export TMPDIR=/tmp
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/pmcb_${tag}_a -- python3 $root/tools/blur_run.py "$@" > $out/pmcb_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d $out/pmcb_${tag}_b -- python3 $root/tools/blur_run.py "$@" > $out/pmcb_${tag}_b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d $out/pmcb_${tag}_c -- python3 $root/tools/blur_run.py "$@" > $out/pmcb_${tag}_c.log 2>&1
python3 $root/tools/pmc_kernel.py blur $out/pmcb_${tag}_a $out/pmcb_${tag}_b $out/pmcb_${tag}_c > $out/pmcb_${tag}.json
cat $out/pmcb_${tag}.json
