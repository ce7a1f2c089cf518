set -e
out=gpurun_out/r05e; mkdir -p $out
python tools/ab_equal.py neilpy_amd/_lib/variants/base.so > $out/ab_equal.log 2>&1 || { tail -20 $out/ab_equal.log; exit 1; }
tail -1 $out/ab_equal.log
python tools/soak.py --reps 300 > $out/soak_f32.log 2>&1; tail -1 $out/soak_f32.log
python tools/soak.py --kind lsqr --size 4097 --holes 0.7 --reps 100 > $out/soak_lsqr.log 2>&1; tail -1 $out/soak_lsqr.log
python tools/soak.py --kind lsqr --size 2049 --holes 0.2 --reps 100 > $out/soak_lsqr_sparse.log 2>&1; tail -1 $out/soak_lsqr_sparse.log
python tools/fuzz_campaign.py --cases 4000 --seed 6101 --kinds smrf,smrf,pf,inpaint,inpaint,dem > $out/fuzz_default_seed6101.log 2>&1; tail -1 $out/fuzz_default_seed6101.log
SMRF_FUSED=2 SMRF_NT=1 SMRF_RING_DUAL=1 python tools/fuzz_campaign.py --cases 4000 --seed 6102 --kinds smrf,pf,pf,pf,inpaint > $out/fuzz_forced_seed6102.log 2>&1; tail -1 $out/fuzz_forced_seed6102.log
