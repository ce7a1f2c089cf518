set -e
out=gpurun_out/r05c; mkdir -p $out
python tools/ab_equal.py neilpy_amd/_lib/variants/base.so > $out/ab_equal.log 2>&1 || { tail -20 $out/ab_equal.log; exit 1; }
tail -2 $out/ab_equal.log
python tools/lsqr_ab.py --size 8193 --occupancy 0.26 --reps 3 --libs neilpy_amd/_lib/variants/base.so > $out/lsqr_ab_random.log 2>&1; tail -3 $out/lsqr_ab_random.log
python tools/lsqr_ab.py --size 8193 --pattern objects --reps 3 --libs neilpy_amd/_lib/variants/base.so > $out/lsqr_ab_objects.log 2>&1; tail -3 $out/lsqr_ab_objects.log
python tools/lsqr_ab.py --size 2000 --occupancy 0.5 --reps 2 --libs neilpy_amd/_lib/variants/base.so > $out/lsqr_ab_small.log 2>&1; tail -3 $out/lsqr_ab_small.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inpaint or smrf or springs" > $out/pytest_lsqr.log 2>&1 || { tail -30 $out/pytest_lsqr.log; exit 1; }
tail -2 $out/pytest_lsqr.log
python tools/window_ab.py --libs neilpy_amd/_lib/variants/base.so --shapes 16384x16384 --windows 50 --reps 5 > $out/ab_16384.log 2>&1
tail -55 $out/ab_16384.log
python tools/window_ab.py --libs neilpy_amd/_lib/variants/base.so --shapes 2048x16384,4096x4096 --windows 50 --reps 5 > $out/ab_small.log 2>&1
grep -E "==|sum" $out/ab_small.log
