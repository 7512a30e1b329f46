# Per-shape FETCH_SIZE / WRITE_SIZE of the GEMM launches (each counter its own rocprofv3 pass, no tracing domains beside --kernel-trace).
# Usage (repo root, GPU box): bash tools/profile_gemm_shapes.sh  -> gpurun_out/prof_gemm_shapes/pmc_gemm_shapes.txt
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_gemm_shapes; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf --output-format csv -- python3 $R/tools/gemm_bench.py > $O/pf.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw --output-format csv -- python3 $R/tools/gemm_bench.py > $O/pw.log 2>&1
python3 $R/tools/pmc_shape_summary.py $(ls $O/pf/*counter_collection.csv $O/pf/*/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/pw/*counter_collection.csv $O/pw/*/*counter_collection.csv 2>/dev/null | head -1) > $O/pmc_gemm_shapes.txt 2>&1
rm -rf $O/pf $O/pw
cat $O/pmc_gemm_shapes.txt
