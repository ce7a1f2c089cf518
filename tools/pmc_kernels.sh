#!/bin/bash
# SQ counters of every kernel of one progressive_filter call (developer tool, run on the GPU box):
#   bash tools/pmc_kernels.sh <out_dir> [--size 16384 --windows 50]
# Three rocprofv3 --pmc passes of tools/pmc_traffic.py (counters only, --kernel-trace; no other trace domain), then
# tools/pmc_kernels.py reduces them to one line per kernel instance.
set -e
OUT=$(readlink -f $1); shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/tools/pmc_traffic.py "$@" > $OUT/pmc$i.log 2>&1 < /dev/null
done
python3 $R/tools/pmc_kernels.py $OUT > $OUT/summary.md
