#!/bin/bash
# Regenerate the measurements kept under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r02
# bench.py default run, its rocprofv3 kernel trace, the two PMC passes (separate runs, no other
# trace domain), and the 4096^2 / 18-window secondary configuration.
set -e
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --size 4096 --windows 18 --steps 20 --warmup 2 --cpu-crop 0 > $out/bench_4096_w18.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 3 --no-cpu > $out/bench_under_rocprof.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 tools/pmc_traffic.py > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 tools/pmc_traffic.py > /dev/null 2> $out/pmc_write.err
tools/ubench/op_rate $out/op_rate_table.md > $out/op_rate.log 2>&1
tools/ubench/issue_rate $out/issue_rate_table.md > $out/issue_rate.log 2>&1
tools/ubench/mix_rate $out/mix_rate_table.md > $out/mix_rate.log 2>&1
ls $out/trace/*/ | head
