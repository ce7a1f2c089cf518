#!/bin/bash
# HBM traffic of the fp64 progressive_filter (the dtype neilpy.smrf runs, neilpy.py:1136) from the hardware counters:
#   bash tools/pmc_f64.sh <out_dir> [--size 8192 --windows 18]
# Two rocprofv3 --pmc passes of tools/pmc_traffic.py --dtype f64 (counters only, --kernel-trace), reduced by tools/pmc_f64.py.
set -e
OUT=$(readlink -f $1); shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/tools/pmc_traffic.py --dtype f64 "$@" > $OUT/$c.log 2>&1 < /dev/null
done
python3 $R/tools/pmc_f64.py $OUT "$@" > $OUT/summary.md
cat $OUT/summary.md
