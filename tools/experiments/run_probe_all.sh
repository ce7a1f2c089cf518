# per-radius A/B (erosion and dilation + flag), every radius: bash tools/experiments/run_probe_all.sh <variant.so> <tag> [f64]
set -e
V=$1; T=$2
mkdir -p gpurun_out/s2
R=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64
python tools/ab_equal.py $V > gpurun_out/s2/${T}_equal.log 2>&1 < /dev/null
python tools/ring_probe.py --radii $R --reps 7 --libs $V > gpurun_out/s2/${T}_all_erode.log 2>&1 < /dev/null
python tools/ring_probe.py --radii $R --reps 7 --flag --libs $V > gpurun_out/s2/${T}_all_flag.log 2>&1 < /dev/null
python tools/ring_probe.py --n 8192 --dtype f64 --radii $R --reps 7 --libs $V > gpurun_out/s2/${T}_f64all_erode.log 2>&1 < /dev/null
python tools/ring_probe.py --n 8192 --dtype f64 --radii $R --reps 7 --flag --libs $V > gpurun_out/s2/${T}_f64all_flag.log 2>&1 < /dev/null
tail -n 2 gpurun_out/s2/${T}_equal.log
