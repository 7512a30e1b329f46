"""MI355X-native `src/clip_whisper` hot path: the reference package's public names (src/clip_whisper/__init__.py:2-12).

    from avllm import ClipWhisperModel, ModalityConnector, ClipWhisperTrainer, AVSRDataset, create_dataloaders

Resolved lazily: importing the package does not import torch or load libavllm.so."""

_EXPORTS = {
    "ClipWhisperModel": "model", "ModalityConnector": "connector", "create_modality_connector": "connector",
    "ClipWhisperTrainer": "trainer", "AVSRDataset": "data", "create_dataloaders": "data",
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        import importlib
        return getattr(importlib.import_module("." + _EXPORTS[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
