#!/bin/bash
# Persistent GEMM tile-boundary experiments (round 2): where do the ~7 us per tile outside the K loop go?  Every variant in ONE run: boxes
# differ by +-10 %.  AVLLM_GEMM_DBG: bit 0 = epilogue without its global stores, bit 1 = strict first-K-step wait (stores not left in
# flight), bits 4.. = number of start-stagger phases (1 = off, 0 = automatic: 2 when the grid walks >= 6 rounds)
cd "$(dirname "$0")/.."
# the knob only exists in an experiment build of the library (the shipped build ignores AVLLM_GEMM_DBG)
AVLLM_EXTRA_FLAGS=-DAVLLM_EXPERIMENT_KNOBS python build.py --force >/dev/null || exit 1
for dbg in ${DBGS:-0 16 2 18 1}; do
  echo "== AVLLM_GEMM_DBG=$dbg"
  AVLLM_GEMM_DBG=$dbg ROWS=4096 python tools/gemm_bench.py 2>/dev/null | grep -E "clip|llama q/k|llama down |whisper qkv"
done
