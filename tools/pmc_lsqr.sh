#!/bin/bash
# HBM traffic of the spring-inpaint LSQR kernels from the hardware counters (run on the GPU box):
#   bash tools/pmc_lsqr.sh <out_dir> [--size 8193 --holes 0.74]
# Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, --kernel-trace), then tools/pmc_lsqr.py.
set -e
OUT=$(readlink -f $1); shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/tools/pmc_lsqr_run.py "$@" > $OUT/$c.log 2>&1 < /dev/null
done
python3 $R/tools/pmc_lsqr.py $OUT > $OUT/summary.md
cat $OUT/summary.md
