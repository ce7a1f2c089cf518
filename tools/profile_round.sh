#!/bin/bash
# Regenerate the measurements kept under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r04
# bench.py default run and the 4096^2 / 18-window configuration; the rocprofv3 kernel trace of the benchmark (no PMC child
# runs inside a traced run: --no-pmc), the two HBM-traffic PMC passes (separate runs, no other trace domain), the SQ
# counters of every kernel of one call, kernel traces of the full smrf() on 20 M and 100 M points (the LSQR kernels) and of
# the fp64 progressive_filter, the compute-only time of a 1/8 band, the misc op-rate micro-benchmark.
# Under rocprofv3 the program after "--" is always python3 / a binary itself (no env, no bash -c).
set -e
tag=${1:-r05}
out=$(pwd)/gpurun_out/$tag
R=$(pwd)
mkdir -p $out
export TMPDIR=/tmp
part=${2:-all}          # A: the benchmark's own files; B: smrf / fp64 / band / micro-benchmarks / LSQR; all: both
if [ "$part" != "B" ]; then
python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
python bench.py --size 4096 --windows 18 --steps 20 --warmup 2 --cpu-crop 0 --no-pmc --no-secondary > $out/bench_4096_w18.json 2>> $out/bench.err < /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --steps 3 --no-cpu --no-pmc --no-secondary --no-power > $out/bench_under_rocprof.json 2> $out/trace.err < /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $R/tools/pmc_traffic.py > /dev/null 2> $out/pmc_fetch.err < /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $R/tools/pmc_traffic.py > /dev/null 2> $out/pmc_write.err < /dev/null
cd $R
bash tools/pmc_kernels.sh $out/pmc_kernels --size 16384 --windows 50
fi
if [ "$part" = "A" ]; then exit 0; fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_smrf20 -- python3 $R/tools/smrf_stages.py --points 20000000 --extent 8192 > $out/smrf_stages_20M.json 2> $out/smrf20.err < /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_smrf100 -- python3 $R/tools/smrf_stages.py --points 100000000 --extent 32768 > $out/smrf_stages_100M.json 2> $out/smrf100.err < /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_f64 -- python3 $R/tools/window_ab.py --shapes 8192x8192 --windows 50 --dtype f64 --fused 0 --reps 3 > $out/f64_windows.log 2> $out/f64.err < /dev/null
cd $R
python tools/band_compute.py --reps 5 > $out/band_compute.log 2>&1 < /dev/null
for w in 2 4; do python tools/band_compute.py --world $w --rank $((w/2)) --reps 4 2>&1 | grep -E "budget (0|None), overlap False" >> $out/band_compute.log; done
# the micro-benchmarks are built from their sources here (no binary is tracked: ADVICE r4)
for u in misc_rate stream_rate; do
  hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/$u
done
tools/ubench/misc_rate $out/misc_rate.md > $out/misc_rate.log 2>&1
tools/ubench/stream_rate $out/stream_rate.md > $out/stream_rate.log 2>&1
# round 4: the LSQR kernels against the counters (FETCH_SIZE / WRITE_SIZE passes) and a kernel trace of one solve
bash tools/pmc_lsqr.sh $out/pmc_lsqr > $out/pmc_lsqr.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_lsqr -- python3 $R/tools/pmc_lsqr_run.py > $out/lsqr_trace.log 2> $out/lsqr_trace.err < /dev/null
cd $R
# round 5: where the time of one solve goes (timestamps of the trace above)
python tools/lsqr_attribution.py $(find $out/trace_lsqr -name "*kernel_trace.csv" | head -1) $out/lsqr_attribution.md > /dev/null
find $out -name "*kernel_stats.csv" | head
