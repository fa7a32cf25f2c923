"""Geometry prefetch for the PointNet++ set-abstraction chain (inference).

Furthest-point sampling and the ball queries of a PVCNN2 U-Net depend on the point COORDINATES only
(reference models/pvcnn2_ada.py:354-372: ``furthest_point_sample(coords, num_centers)`` feeds the next
stage's coords), never on the features, yet in the module order they sit behind the stage's voxel
convolutions.  FPS is a latency-bound serial loop (1024 rounds on one workgroup per cloud, 32 CUs
busy); run in line it costs ~1 ms of a 22 ms denoiser step with 7/8 of the chip idle.  ``prefetch``
issues the whole chain -- FPS 2048->1024->256->64->16 and every stage's ball query -- on a side HIP
stream at the start of the forward pass, where it overlaps the first PVConv's MFMA convolutions;
``furthest_point_sample`` / ``ball_query`` then pick the results up (event wait, no recomputation).
Inside a hipGraph capture the side stream becomes a parallel branch of the graph.
The lookup is keyed by tensor identity; a miss simply computes in line, so results never change.
"""
import contextlib

import torch

_PLAN = None
_SIDE = {}
# Off by default (round 3): inside a captured step the side stream becomes a parallel graph branch, and hipGraph replays
# of multi-branch graphs are SLOWER on ROCm 7.2 than the same kernels on one stream -- B = 32 sampling step, same box,
# `python bench.py --steps 20 --warmup 5`: prefetch + point-branch streams 13.1-13.2 ms, prefetch only 14.7, point branch
# only 11.1, single stream 10.6 (profiles/archive/r03_streams_and_graph_knobs.txt; DEBUG_CLR_GRAPH_PACKET_CAPTURE, graph queue
# count and CP-wait knobs do not change it; under rocprofv3 the branches do overlap and the step takes 9.1 ms).  Eager
# launches gain 0.4 ms from the prefetch (9.2 -> 8.8 ms).  LION_GEOMETRY_PREFETCH=1 turns it on.
ENABLED = __import__("os").environ.get("LION_GEOMETRY_PREFETCH", "0") != "0"
# What does overlap: SEPARATE single-branch graphs on separate streams.  A captured sampling chain (lion_amd/chain.py) runs the
# geometry chain of a step as its own graph on the geometry stream, beside the part of the forward that does not need it
# yet (compute_chain / external below).  LION_GEOMETRY_SPLIT_GRAPH=0 keeps everything in one graph on one stream.
SPLIT_GRAPH = __import__("os").environ.get("LION_GEOMETRY_SPLIT_GRAPH", "1") != "0"


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        # high priority: the chain is latency bound (1356 serial FPS rounds) and must not queue behind the 400-us
        # convolution workgroups it overlaps with
        _SIDE[key] = torch.cuda.Stream(device=device, priority=-1)
    return _SIDE[key]


def _take(entry):
    tensor, event = entry[-2], entry[-1]
    if not isinstance(event, torch.cuda.Event):   # an entry of an external plan (split-graph chains): `event` is the stage
        plan = _PLAN                              # index; ordering is the plan owner's business
        if plan is not None and plan.get("on_use") is not None and event is not None and event > plan["seen"]:
            first, last = plan["seen"] + 1, event
            plan["seen"] = event
            plan["on_use"](first, last)           # stages first .. last are about to be read for the first time
        return tensor
    cur = torch.cuda.current_stream(tensor.device)
    cur.wait_event(event)
    tensor.record_stream(cur)
    return tensor


def lookup_fps(coords, num_samples):
    if _PLAN is None:
        return None
    hit = _PLAN["fps"].get((id(coords), int(num_samples)))
    return None if hit is None else _take(hit)


def lookup_ball_query(centers, points, radius, num_neighbors):
    if _PLAN is None:
        return None
    hit = _PLAN["bq"].get((id(centers), id(points), float(radius), int(num_neighbors)))
    return None if hit is None else _take(hit)


def compute_chain(sa_modules, coords, stream=None, boundary=None):
    """FPS -> ball queries -> next FPS ... of the set-abstraction stages on `coords`, launched on the current stream.
    With `stream` given an event is recorded behind every result (the in-forward prefetch); without, entries carry
    their stage index (external plans: a chain runner orders whole graphs instead) and `boundary(i)` is called between
    stage i - 1 and stage i (the runner cuts its geometry graph there)."""
    from .functional.ball_query import _ball_query_compute
    from .functional.sampling import _fps_compute
    plan = {"fps": {}, "bq": {}, "root": coords, "stages": 0}
    cur = coords
    for i, m in enumerate(sa_modules):
        if getattr(m, "num_centers", None) is None:
            break
        if boundary is not None and i > 0:
            boundary(i)
        centers = _fps_compute(cur, m.num_centers)
        ev = i
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
        plan["fps"][(id(cur), int(m.num_centers))] = (cur, centers, ev)  # `cur` kept alive: its id is the key
        for g in m.groupers:
            idx = _ball_query_compute(centers, cur, g.radius, g.num_neighbors)
            ev2 = i
            if stream is not None:
                ev2 = torch.cuda.Event()
                ev2.record(stream)
            plan["bq"][(id(centers), id(cur), float(g.radius), int(g.num_neighbors))] = (centers, cur, idx, ev2)
        cur = centers
        plan["stages"] = i + 1
    return plan


_EXTERNAL = None


@contextlib.contextmanager
def external(plan, on_use=None):
    """A chain runner (lion_amd/chain.py, split-graph mode) computed `plan` = compute_chain(...) on a static copy of the
    coordinates the forward inside this context will see, in its own graph(s) on its own stream.  The forward's
    `prefetch()` then adopts it instead of launching anything; `on_use(first, last)` is called right before a result of a
    stage not seen so far is handed to the forward (the runner cuts its main graph / waits for the geometry stream)."""
    global _EXTERNAL
    prev, _EXTERNAL = _EXTERNAL, (plan, on_use)
    try:
        yield
    finally:
        _EXTERNAL = prev


def _adopt(ext, coords):
    """the external plan, re-keyed on this forward's root tensor (every other key is a tensor of the plan itself)"""
    plan, on_use = ext
    root = plan["root"]
    if tuple(root.shape) != tuple(coords.shape) or root.device != coords.device:
        return None
    out = {"fps": dict(plan["fps"]), "bq": dict(plan["bq"]), "on_use": on_use, "seen": -1, "keep": coords}
    for (cid, n), ent in plan["fps"].items():
        if cid == id(root):
            out["fps"][(id(coords), n)] = ent
    for (ctr, pts, r, k), ent in plan["bq"].items():
        if pts == id(root):
            out["bq"][(ctr, id(coords), r, k)] = ent
    return out


@contextlib.contextmanager
def prefetch(sa_modules, coords):
    """sa_modules: the PointNetSAModule of each stage, in order; coords f32[B,3,N] (the tensor object
    the first stage will receive)."""
    global _PLAN
    if _EXTERNAL is not None and sa_modules and coords.is_cuda and not torch.is_grad_enabled():
        adopted = _adopt(_EXTERNAL, coords)
        if adopted is not None:
            prev, _PLAN = _PLAN, adopted
            try:
                yield
            finally:
                _PLAN = prev
            return
    if (not ENABLED or not sa_modules or not coords.is_cuda or torch.is_grad_enabled() or coords.dim() != 3
            or coords.shape[1] != 3 or not coords.is_contiguous() or coords.dtype != torch.float32):
        yield
        return
    main = torch.cuda.current_stream(coords.device)
    side = _side_stream(coords.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        plan = compute_chain(sa_modules, coords, side)
    prev, _PLAN = _PLAN, plan
    try:
        yield
    finally:
        _PLAN = prev
        main.wait_stream(side)  # join (closes the branch inside a graph capture; harmless otherwise)


# ---- sharing between networks that see the same cloud ------------------------------------------------------------------------
# The VAE's two encoders (style encoder: models/shapelatent_modules.py, latent-point encoder: models/latent_points_ada.py;
# reference models/vae_adain.py:92-118) both start with set abstractions (1024 centres, radius 0.1, 32 neighbours) and (256, 0.2, 32)
# on the SAME input cloud: the furthest-point samples and the ball queries of those two stages are computed twice, and FPS is a
# serial loop (547 + 111 us at B = 32, one workgroup per cloud).  Inside `shared()` the results are memoised on the identity of
# their inputs -- (address, shape, strides, version counter) with the input kept alive by the entry -- so the second network picks
# up the first one's tensors.  Only around code that does not write those inputs through raw pointers (the version counter sees
# torch's in-place ops only): the context is entered by the VAE's encode, nowhere else.
_MEMO = None
SHARED = __import__("os").environ.get("LION_GEOMETRY_SHARED", "1") != "0"


def _ident(t):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, t.dtype)


@contextlib.contextmanager
def shared():
    global _MEMO
    if not SHARED:
        yield
        return
    prev, _MEMO = _MEMO, ({} if _MEMO is None else _MEMO)
    try:
        yield
    finally:
        _MEMO = prev


def memo_get(kind, tensors, *scalars):
    if _MEMO is None:
        return None, None
    key = (kind,) + tuple(_ident(t) for t in tensors) + tuple(scalars)
    hit = _MEMO.get(key)
    return key, (None if hit is None else hit[1])


def memo_put(key, tensors, value):
    if _MEMO is not None and key is not None:
        _MEMO[key] = (tuple(tensors), value)   # the inputs stay alive: their addresses cannot be handed to other tensors
    return value
