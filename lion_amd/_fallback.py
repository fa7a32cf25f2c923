"""Accounting of every place where a layer leaves this library's kernels for a vendor-library / ATen implementation
(MIOpen convolution, rocBLAS matmul, ATen elementwise walks).  Such paths exist for shapes and modes the HIP kernels do
not cover (autocast, odd channel counts, resolutions other than 8 / 16 / 32); they are never a CPU path and never the
oracle.  Each one goes through `note()`:

  * `counts()` says how often each was taken (the benchmark and the B = 32 replay test assert it stays empty for the
    sampling step they measure);
  * with LION_STRICT=1 in the environment (or `strict(True)`) the first one raises instead -- a run that must stay on
    the hand-written kernels fails loudly rather than silently timing the vendor library."""
import os
from collections import Counter

_COUNTS = Counter()
_REASONS = Counter()
_STRICT = os.environ.get("LION_STRICT", "0") not in ("", "0")


def strict(on=None) -> bool:
    global _STRICT
    if on is not None:
        _STRICT = bool(on)
    return _STRICT


def note(site: str, reason: str = "") -> None:
    _COUNTS[site] += 1
    if reason:
        _REASONS[f"{site}: {reason}"] += 1
    if _STRICT:
        raise RuntimeError(f"LION_STRICT: {site} left the HIP kernels for the vendor library" + (f" ({reason})" if reason else ""))


def counts() -> dict:
    return dict(_COUNTS)


def reasons(top: int = 20) -> dict:
    """the most frequent (site: reason) pairs -- which shapes left the library's kernels"""
    return dict(_REASONS.most_common(top))


def reset() -> None:
    _COUNTS.clear()
    _REASONS.clear()
