"""Word error rate as scripts/clip_whisper/decode.py:30-37 computes it (`jiwer.wer(refs, hyps)`): corpus-level
(S+D+I)/N over whitespace-split words, raw strings, no normalisation; `inf` when it cannot be computed.
jiwer itself is not installed offline: restated from its definition, pinned by hand-computed cases in tests."""
from __future__ import annotations


def _edit_distance(ref, hyp):
    prev = list(range(len(hyp) + 1))
    for i in range(1, len(ref) + 1):
        cur = [i] + [0] * len(hyp)
        for j in range(1, len(hyp) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ref[i - 1] != hyp[j - 1]))
        prev = cur
    return prev[len(hyp)]


def calculate_wer(references, hypotheses):
    try:
        if isinstance(references, str):
            references, hypotheses = [references], [hypotheses]
        errs = sum(_edit_distance(r.split(), h.split()) for r, h in zip(references, hypotheses))
        n = sum(len(r.split()) for r in references)
        return errs / n if n else float("inf")
    except Exception:
        return float("inf")
