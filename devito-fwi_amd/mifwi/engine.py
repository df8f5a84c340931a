"""
Plan cache between the Python solver objects and libmifwi.

The reference builds a new `AcousticWaveSolver` (hence new Devito operators) for every shot of
every objective evaluation (fwi.py:62,134).  Here solver objects are cheap views: GPU plans are
keyed by the discretisation and reused, so device buffers, geometry tables and captured graphs
survive across shots, evaluations and solver instances.
"""
import collections

import numpy as np

from ._lib import Plan


class PlanEntry(object):
    """A cached plan plus bookkeeping of which wavefields it currently holds."""

    def __init__(self, plan):
        self.plan = plan
        self._tokens = collections.defaultdict(int)   # wavefield storage id (mifwi_read_wavefield `which`) -> generation
        self._live = {}
        self.model_key = None
        self.geom_key = None

    def token(self, which):
        return self._tokens[which]

    def claim(self, which, owner=None):
        """Called before a run overwrites wavefield storage `which`: a still-unread lazy wavefield
        that refers to it is downloaded first, then the generation is bumped."""
        import weakref
        ref = self._live.get(which)
        old = ref() if ref is not None else None
        if old is not None and old.on_device():
            old.materialize()
        self._tokens[which] += 1
        self._live[which] = weakref.ref(owner) if owner is not None else None
        return self._tokens[which]

    def register(self, which, owner):
        import weakref
        self._live[which] = weakref.ref(owner)


_CACHE = collections.OrderedDict()
_MAX_PLANS = 4


def _key(shape, spacing, origin, space_order, nt, dt, nsrc, nrec, max_shots, dtype, device, flags=0):
    return (tuple(int(s) for s in shape), tuple(float(s) for s in spacing), tuple(float(o) for o in origin),
            int(space_order), int(nt), float(dt), int(nsrc), int(nrec), int(max_shots), np.dtype(dtype).str,
            int(device), int(flags))


def get_plan(shape, spacing, origin, space_order, nt, dt, nsrc, nrec, max_shots=1, dtype=np.float32,
             device=-1, flags=0):
    """Return a cached PlanEntry for this discretisation, creating (and evicting) as needed."""
    key = _key(shape, spacing, origin, space_order, nt, dt, nsrc, nrec, max_shots, dtype, device, flags)
    entry = _CACHE.get(key)
    if entry is not None:
        _CACHE.move_to_end(key)
        return entry
    while len(_CACHE) >= _MAX_PLANS:
        _, old = _CACHE.popitem(last=False)
        for which in (0, 1, 2):
            old.claim(which)          # download anything still referenced lazily
        old.plan.close()
    entry = PlanEntry(Plan(shape, spacing, origin, space_order, nt, dt, nsrc, nrec, max_shots=max_shots,
                           dtype=dtype, device=device, flags=flags))
    _CACHE[key] = entry
    return entry


def clear():
    while _CACHE:
        _, old = _CACHE.popitem(last=False)
        for which in (0, 1, 2):
            old.claim(which)
        old.plan.close()
