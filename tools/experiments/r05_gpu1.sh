set -e
out=gpurun_out/r05a; mkdir -p $out
R=$(pwd)
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err
python -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], json.dumps(d['roofline'].get('secondary_bound'))); print(json.dumps(d['roofline']['classes']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace_lsqr -- python3 $R/tools/pmc_lsqr_run.py > $R/$out/lsqr_trace.log 2> $R/$out/lsqr_trace.err
cd $R
python tools/lsqr_attribution.py $(find $out/trace_lsqr -name "*kernel_trace.csv") $out/lsqr_attribution.md
tail -1 $out/lsqr_trace.log
python tools/pmc_lsqr_run.py > $out/lsqr_noprof.log 2>&1; tail -1 $out/lsqr_noprof.log
for w in 2 4 8; do python tools/band_compute.py --world $w --rank $((w/2)) --reps 4 2>&1 | grep -E "budget (0|None), overlap False" ; done | tee $out/band_compute.log
