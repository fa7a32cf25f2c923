"""Cache of tensors derived from a weight (packed / mirrored / summed copies).

Keyed by the weight's storage address and shape, validated by its version counter (in-place optimizer steps bump
it).  Every entry holds a STRONG reference to the weight tensor it was built from: while the entry lives the storage
cannot be freed, so its address cannot be handed to a different tensor -- an id()- or address-keyed cache without
that reference returns the previous owner's packed copy when a freed parameter's block is reused by a new parameter
of the same shape (seen as a flaky parity test).  Least-recently-used entries are dropped beyond `capacity`.

The version counter only sees writes made through the tensor itself: ``p.data.copy_()`` / ``p.data[...] = `` write
the same storage WITHOUT bumping it.  Writers inside this package therefore use ``with torch.no_grad(): p.copy_()``
and additionally call ``invalidate_all()``, which advances a process-wide generation every entry (and every derived
object: ``StylePlan``, ``GraphedDenoiser``) is validated against; code outside the package that pokes ``.data`` must
call ``lion_amd.invalidate_weight_caches()`` itself."""
from collections import OrderedDict
from contextlib import contextmanager

_GENERATION = 0
_PIN_LISTS = []   # active `pinning()` lists: every cache value handed out meanwhile is appended to each of them


@contextmanager
def pinning(keep: list):
    """While active, every value a WeightCache hands out is also appended to `keep`.  A captured graph bakes in raw
    pointers to the packed / mirrored copies its warm-up and capture passes obtained here; holding `keep` for the graph's
    lifetime keeps that memory alive (and its addresses unreusable) even after the LRU has evicted the entries."""
    _PIN_LISTS.append(keep)
    try:
        yield keep
    finally:
        _PIN_LISTS.remove(keep)


def generation() -> int:
    return _GENERATION


def invalidate_all() -> None:
    """Declare every tensor derived from a weight stale (packed / mirrored copies, style plans, captured graphs)."""
    global _GENERATION
    _GENERATION += 1


def fingerprint(module) -> tuple:
    """What a captured graph or a plan over `module`'s weights is valid for: the generation, and the identity and
    version of every parameter and buffer."""
    items = list(module.parameters()) + list(module.buffers())
    return (_GENERATION, tuple((t.data_ptr(), t._version) for t in items))


class WeightCache:
    def __init__(self, build, capacity=256):
        self._build = build
        self._entries = OrderedDict()
        self._capacity = capacity

    def get(self, weight):
        key = (weight.data_ptr(), tuple(weight.shape), weight.dtype, weight.device)
        hit = self._entries.get(key)
        if hit is not None and hit[0] == (weight._version, _GENERATION):
            self._entries.move_to_end(key)
            for keep in _PIN_LISTS:
                keep.append(hit[2])
            return hit[2]
        value = self._build(weight)
        for keep in _PIN_LISTS:
            keep.append(value)
        self._entries[key] = ((weight._version, _GENERATION), weight, value)
        self._entries.move_to_end(key)
        while len(self._entries) > self._capacity:
            self._entries.popitem(last=False)
        return value

    def clear(self):
        self._entries.clear()
