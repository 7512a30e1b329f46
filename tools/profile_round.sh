# Round-end profile set on one MI355X: kernel trace + stats, FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (each its own run), then a plain bench.
# Usage (from the repo root on the GPU box): bash tools/profile_round.sh  -> gpurun_out/prof_r03/
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${ROUND:-r03}; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inputs --no-decode --no-config5 > $O/kt.log 2>&1
python3 $R/tools/prof_summary.py $(ls $O/kt/*kernel_trace.csv $O/kt/*/*kernel_trace.csv 2>/dev/null | head -1) 0.0 > $O/kernel_trace_summary.txt 2>&1
cp $(ls $O/kt/*kernel_stats.csv $O/kt/*/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-inputs --no-decode --no-config5 > $O/pf.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-inputs --no-decode --no-config5 > $O/pw.log 2>&1
python3 $R/tools/pmc_summary.py $(ls $O/pf/*counter_collection.csv $O/pf/*/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/pw/*counter_collection.csv $O/pw/*/*counter_collection.csv 2>/dev/null | head -1) > $O/pmc_hbm_summary.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pm -o pm --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-inputs --no-decode --no-config5 > $O/pm.log 2>&1
python3 $R/tools/pmc_mfma_summary.py $(ls $O/pm/*counter_collection.csv $O/pm/*/*counter_collection.csv 2>/dev/null | head -1) > $O/pmc_mfma_summary.txt 2>&1
rm -rf $O/kt $O/pf $O/pw $O/pm
cd $R && timeout -k 10 300 python3 bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-400
