#!/bin/bash
# Persistent GEMM tile-boundary experiments (round 2): where do the ~7 us per tile outside the K loop go?
#   AVLLM_GEMM_DBG bit 0: epilogue without its global stores; bits 4..: de-phase the workgroups' start in N phases
cd "$(dirname "$0")/.."
for dbg in 0 1 32 64 128; do
  echo "== AVLLM_GEMM_DBG=$dbg"
  AVLLM_GEMM_DBG=$dbg ROWS=4096 python tools/gemm_bench.py 2>/dev/null | grep -E "clip|llama q/k|llama down "
done
