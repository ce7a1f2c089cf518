set -e
out=gpurun_out/r05d; mkdir -p $out
python tools/ab_equal.py neilpy_amd/_lib/variants/base.so > $out/ab_equal.log 2>&1 || { tail -20 $out/ab_equal.log; exit 1; }
tail -1 $out/ab_equal.log
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
python tools/window_ab.py --libs neilpy_amd/_lib/variants/base.so --shapes 16384x16384 --windows 50 --first 15 --reps 6 --fused 0 > $out/ab_16384.log 2>&1
tail -40 $out/ab_16384.log
python bench.py > $out/bench.json 2> $out/bench.err
python -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], json.dumps(d['roofline'].get('secondary_bound'))); s=d['secondary']['smrf_20M']; print(s['smrf_total_ms'], s['inpaint1_ms'], s['inpaint1'], s['inpaint2_ms'], s['inpaint2']); print(d['secondary']['progressive_filter_f64_8192_w18']['ms_per_step'])"
