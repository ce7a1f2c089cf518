# A/B of the current library against a variant build on one box: bash tools/experiments/run_ab.sh <variant.so> <tag>
set -e
V=$1; T=$2
mkdir -p gpurun_out/s2
python tools/ab_equal.py $V > gpurun_out/s2/${T}_equal.log 2>&1 < /dev/null
python tools/ring_probe.py --radii 9,12,16,20,24,28,32,33,36,40,44,50 --reps 5 --libs $V > gpurun_out/s2/${T}_probe_erode.log 2>&1 < /dev/null
python tools/ring_probe.py --radii 16,24,32,40,50 --reps 5 --flag --libs $V > gpurun_out/s2/${T}_probe_flag.log 2>&1 < /dev/null
python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_cur.json 2>gpurun_out/s2/${T}_bench.err < /dev/null
NEILPY_AMD_LIB=$V python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_var.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_cur2.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
tail -n 1 gpurun_out/s2/${T}_equal.log
grep -v amdgpu.ids gpurun_out/s2/${T}_probe_erode.log
python -c "
import json
for f in ('cur','var','cur2'):
    d=json.load(open('gpurun_out/s2/${T}_bench_%s.json'%f)); print(f, round(d['ms_per_step'],3))
"
