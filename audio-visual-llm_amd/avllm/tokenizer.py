"""Tokenizer plumbing.  A local HF tokenizer is used when `llm_path` is a directory that has one; offline
(no checkpoints: SURVEY.md §8c) a byte-level stand-in keeps the CLI/trainer/decode surface runnable."""
from __future__ import annotations

import os
import types

import torch


class ByteTokenizer:
    """ids: 0 unk, 1 bos, 2 eos(=pad), 3.. = byte+3 (vocab must be >= 259)."""
    bos_token_id, eos_token_id, unk_token_id = 1, 2, 0

    def __init__(self, vocab_size):
        self.vocab_size = vocab_size
        self.pad_token_id = 2
        self.pad_token = "</s>"

    def encode(self, text, add_bos=True):
        ids = [b + 3 for b in text.encode("utf-8") if b + 3 < self.vocab_size]
        return ([1] if add_bos else []) + ids

    def __call__(self, texts, return_tensors=None, padding=False, truncation=False, max_length=None, **kw):
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode(t) for t in texts]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.full((len(rows), width), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros_like(ids)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
            mask[i, : len(r)] = 1
        return types.SimpleNamespace(input_ids=ids, attention_mask=mask)

    def decode(self, ids, skip_special_tokens=True):
        bs = bytes(int(i) - 3 for i in ids if int(i) >= 3 and int(i) - 3 < 256)
        return bs.decode("utf-8", errors="replace")

    def batch_decode(self, batch, skip_special_tokens=True):
        return [self.decode(row, skip_special_tokens) for row in batch]


_TOKENIZER_FILES = ("tokenizer.json", "tokenizer.model", "tokenizer_config.json")


def load_tokenizer(llm_path, vocab_size, synthetic=False):
    """The LLM directory's own tokenizer (errors while loading it propagate: a broken tokenizer must not silently become a byte
    tokenizer).  The byte-level stand-in is used only in synthetic mode (no checkpoints offline) -- otherwise a path without
    tokenizer files raises, and the caller can hand one over as `_provided_tokenizer`."""
    if isinstance(llm_path, str) and os.path.isdir(llm_path) and any(os.path.exists(os.path.join(llm_path, f)) for f in _TOKENIZER_FILES):
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(llm_path, local_files_only=True)
        if tok.pad_token is None:
            tok.pad_token = tok.eos_token
        return tok
    if synthetic:
        return ByteTokenizer(vocab_size)
    raise FileNotFoundError(f"no tokenizer files ({', '.join(_TOKENIZER_FILES)}) under {llm_path!r}: pass _provided_tokenizer, or "
                            "synthetic_weights=True for the byte-level stand-in")
