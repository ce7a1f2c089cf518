set -e
out=gpurun_out/r05b; mkdir -p $out
python -m pytest tests/test_gpu_w50.py -m gpu -x -q > $out/pytest_w50.log 2>&1 || { tail -30 $out/pytest_w50.log; exit 1; }
tail -2 $out/pytest_w50.log
python tools/window_ab.py --libs neilpy_amd/_lib/variants/base.so,neilpy_amd/_lib/variants/fuse.so --shapes 16384x16384 --windows 50 --reps 4 > $out/ab_16384.log 2>&1
tail -60 $out/ab_16384.log
python tools/window_ab.py --libs neilpy_amd/_lib/variants/base.so,neilpy_amd/_lib/variants/fuse.so --shapes 2048x16384,4096x4096 --windows 50 --reps 4 > $out/ab_small.log 2>&1
grep -E "==|sum" $out/ab_small.log
