"""Minimal stand-in so the reference's modules import in the fixture generator (loguru is not installed)."""


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None

    def catch(self, *a, **k):
        def deco(f):
            return f
        return deco if not (a and callable(a[0])) else a[0]


logger = _Logger()
