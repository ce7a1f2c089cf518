# per-radius A/B (erosion and dilation + flag), every ring radius: bash tools/experiments/run_probe_all.sh <variant.so> <tag>
set -e
V=$1; T=$2
mkdir -p gpurun_out/s2
R=9,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50
python tools/ring_probe.py --radii $R --reps 7 --libs $V > gpurun_out/s2/${T}_all_erode.log 2>&1 < /dev/null
python tools/ring_probe.py --radii $R --reps 7 --flag --libs $V > gpurun_out/s2/${T}_all_flag.log 2>&1 < /dev/null
echo done
