# Kernel trace of the BASELINE config 5 variant (fp8): per-kernel time of a step.  -> gpurun_out/prof_c5/
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c5; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --precision fp8 --batch 4 --frames 750 --whisper openai/whisper-large-v3 --clip openai/clip-vit-large-patch14 --llm mistralai/Mistral-7B-v0.1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inputs --no-decode > $O/kt.log 2>&1
python3 $R/tools/prof_summary.py $(ls $O/kt/*kernel_trace.csv $O/kt/*/*kernel_trace.csv 2>/dev/null | head -1) 0.0 > $O/kernel_trace_summary.txt 2>&1
rm -rf $O/kt
head -30 $O/kernel_trace_summary.txt
