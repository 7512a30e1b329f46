# Kernel trace of the greedy-decode bench (tools/decode_bench.py, B=8): per-kernel time of a token step.  -> gpurun_out/prof_decode/
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_decode; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/decode_bench.py --new 32 > $O/kt.log 2>&1
cp $(ls $O/kt/*kernel_stats.csv $O/kt/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
rm -rf $O/kt
head -14 $O/kernel_stats.csv | cut -c1-200
