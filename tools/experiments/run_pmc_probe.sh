# PMC counters of single ring launches: bash tools/experiments/run_pmc_probe.sh <lib.so> <radii> <tag>
set -e
LIBV=$1; RAD=$2; T=$3
OUT=/root/repo/gpurun_out/s2/pmc_$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NEILPY_AMD_LIB=/root/repo/$LIBV
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 /root/repo/tools/ring_probe.py --radii $RAD --reps 1 > $OUT/p1.log 2>&1 < /dev/null
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 /root/repo/tools/ring_probe.py --radii $RAD --reps 1 > $OUT/p2.log 2>&1 < /dev/null
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 /root/repo/tools/ring_probe.py --radii $RAD --reps 1 > $OUT/p3.log 2>&1 < /dev/null
python3 /root/repo/tools/pmc_summary.py $OUT
