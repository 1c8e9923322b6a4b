"""Inert stand-in for `loguru` (off the hot path: one warning at D4:3784)."""
class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None
logger = _Logger()
