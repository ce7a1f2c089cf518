# the measurements kept under profiles/ (the bench / trace / PMC part of tools/profile_round.sh)
set -e
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
python bench.py --size 4096 --windows 18 --steps 20 --warmup 2 --cpu-crop 0 --no-pmc > $out/bench_4096_w18.json 2>> $out/bench.err < /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 3 --no-cpu --no-pmc > $out/bench_under_rocprof.json 2> $out/trace.err < /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 tools/pmc_traffic.py > /dev/null 2> $out/pmc_fetch.err < /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 tools/pmc_traffic.py > /dev/null 2> $out/pmc_write.err < /dev/null
python tools/band_compute.py --reps 5 > $out/band_compute.log 2>&1 < /dev/null
python tools/smrf_stages.py --points 20000000 --extent 8192 > $out/smrf_stages_20M.log 2>&1 < /dev/null
ls $out/trace/*/
