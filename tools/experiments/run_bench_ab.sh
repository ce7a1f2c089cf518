# whole-benchmark A/B of the current library against a variant: bash tools/experiments/run_bench_ab.sh <variant.so> <tag>
set -e
V=$1; T=$2
mkdir -p gpurun_out/s2
python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_cur.json 2>gpurun_out/s2/${T}_bench.err < /dev/null
NEILPY_AMD_LIB=$V python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_var.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_cur2.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
NEILPY_AMD_LIB=$V python bench.py --no-cpu --no-pmc > gpurun_out/s2/${T}_bench_var2.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
NEILPY_AMD_LIB=$V python bench.py --no-cpu --no-pmc --size 4096 --windows 18 --steps 20 > gpurun_out/s2/${T}_bench4096_var.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
python bench.py --no-cpu --no-pmc --size 4096 --windows 18 --steps 20 > gpurun_out/s2/${T}_bench4096_cur.json 2>>gpurun_out/s2/${T}_bench.err < /dev/null
python -c "
import json
for f in ('bench_cur','bench_var','bench_cur2','bench_var2','bench4096_cur','bench4096_var'):
    d=json.load(open('gpurun_out/s2/${T}_%s.json'%f)); print(f, round(d['ms_per_step'],3), round(d['roofline']['frac'],4))
"
