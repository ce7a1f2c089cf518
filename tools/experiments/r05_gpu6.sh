set -e
out=$(pwd)/gpurun_out/r05; R=$(pwd); mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_lsqr -- python3 $R/tools/pmc_lsqr_run.py > $out/lsqr_trace.log 2> $out/lsqr_trace.err < /dev/null
cd $R
python tools/lsqr_attribution.py $(find $out/trace_lsqr -name "*kernel_trace.csv" | head -1) $out/lsqr_attribution.md
tail -1 $out/lsqr_trace.log
python tools/pmc_lsqr_run.py | tail -1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl or bench or sharded" 2>&1 | tail -3
