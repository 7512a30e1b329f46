"""YAML + CLI configuration for the train/decode entry points (same keys/flags as the reference's
configs/clip_whisper.yaml:4-55 and scripts/clip_whisper/train.py:33-81 / decode.py:41-68)."""
from __future__ import annotations

import yaml


def load_config(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def flatten(cfg):
    """nested sections -> one dict; later sections do not override earlier keys of the same name except model > data."""
    out = {}
    for sec in ("data", "processor", "training", "model"):
        for k, v in (cfg.get(sec) or {}).items():
            out[k] = v
    for k, v in cfg.items():
        if not isinstance(v, dict):
            out[k] = v
    return out


def merged(cfg_path, cli: dict):
    cfg = flatten(load_config(cfg_path)) if cfg_path else {}
    for k, v in cli.items():
        if v is not None:
            cfg[k] = v
    return cfg
