"""Whole-step capture of a sampling chain -- what ``DiffusionDiscretized.run_ddim`` /
``run_denoising_diffusion`` / ``LION.sample`` run by default on the GPU.

One denoiser evaluation of the local prior is ~500 kernel launches; the reference's samplers
(utils/diffusion_pvd.py:224-303, :390-473) additionally issue ~10 elementwise launches, a ``torch.full`` and a
host-side ``randn`` + H2D copy per step.  Here a captured step holds
    lion_chain_begin_step  ->  denoiser forward  ->  lion_chain_update_noise
and a chain of S steps is S replays of it: one hipGraph for models without set abstraction (the global prior); for the
point-voxel denoiser three single-branch graphs on the main stream and the step's FPS / ball-query chain as two graphs on a
second stream, ordered by events between the launches (GraphedChain.__init__; round 3 -- branches INSIDE one graph replay
slower than one stream on ROCm 7.2, separate graphs on separate streams overlap).  Everything that changes from step to step lives in device memory:
the schedule table (timestep for the model + the update's coefficients, S x 8 floats uploaded once per chain), the
step counter, the Philox seed, the latent ``x`` (updated in place).  The host issues one to five ``hipGraphLaunch`` per step
and never synchronises inside the chain.

A captured graph bakes in the packed-weight pointers of its model: it is keyed by ``_wcache.fingerprint(model)``
and re-captured when any parameter changed (optimizer step, EMA swap, checkpoint load).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, _wcache

DDIM, DDPM = 0, 1
# tests: a list here makes every chain run append (start latent, [the noise each step drew]) -- what an eager loop needs
# to repeat a graphed chain draw for draw (run_ddim(given_noise=...)); chains are then captured with the noise output on
RECORD = None
TEMB_TABLE = __import__("os").environ.get("LION_TEMB_TABLE", "1") != "0"   # A/B: 0 = every step recomputes its time embedding
CHANNEL_MAJOR_EPS = __import__("os").environ.get("LION_CHAIN_CM_EPS", "1") != "0"   # A/B: 0 = the model transposes its output

def policy_key() -> tuple:
    """The module-level switches that decide WHICH kernels a forward launches: a captured graph is only valid for the
    setting it was captured under (flipping one of them re-captures instead of silently replaying the old launches)."""
    from . import conv_ops, fused_ops, geometry
    from .models import pvcnn2_ada
    return (pvcnn2_ada.SPARSE_CONV1, pvcnn2_ada.FUSE_INFERENCE, pvcnn2_ada.OVERLAP_POINT_BRANCH, geometry.ENABLED,
            geometry.SPLIT_GRAPH, conv_ops.SPLIT, fused_ops.PW_SPLIT, pvcnn2_ada.VOX_PLAN, TEMB_TABLE, fused_ops.MAX_RECOMPUTE,
            pvcnn2_ada.SKIP_UNREAD, pvcnn2_ada.SKIP_UNREAD_LEVEL2, CHANNEL_MAJOR_EPS)


class GraphedChain:
    def __init__(self, model, num_samples, shape, condition_input, clip_feat, device, mode, capacity, warmup=2,
                 record_noise=False):
        self.model, self.mode, self.capacity = model, mode, int(capacity)
        self.fingerprint = _wcache.fingerprint(model)
        dev = torch.device(device)
        size = [num_samples] + list(shape)
        self.x = torch.zeros(size, device=dev)
        self.t = torch.zeros(num_samples, device=dev)
        self.cond = None if condition_input is None else condition_input.detach().clone().contiguous()
        self.clip = None if clip_feat is None else clip_feat.detach().clone().contiguous()
        self.table = torch.zeros(self.capacity, 8, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seed = torch.zeros(2, dtype=torch.int32, device=dev)   # two 32-bit words of the Philox key
        self.cur = torch.zeros(8, device=dev)
        self.z = torch.zeros(size, device=dev) if record_noise else None   # tests: the noise each step used
        # identity rows until run() uploads a schedule: the warm-up / capture passes must run on finite values (an
        # all-zero DDPM row is 0/0 -> the second warm-up forward would voxelize NaN latents)
        self.table[:, 0] = 1.0   # t_model
        self.table[:, 1] = 1.0   # a0: x passes through
        self.table[:, 3] = 1.0   # a2: DDPM divisor (DDIM: z has weight 1 -- finite)
        # The time embedding of a step depends on the step's timestep alone: its rows for the whole chain are computed once
        # per run() (model.time_embedding over the schedule) and a replayed step picks its row by the device-resident step
        # index -- instead of 7 (global prior) / 3 (local prior) launches per step that recompute it for every sample.
        self.temb_table = None
        self.use_temb_table = TEMB_TABLE and hasattr(model, "time_embedding") and not getattr(model, "embed_dim", 1) == 0
        self.policy = policy_key()
        self.pinned = []         # strong references to every packed / mirrored weight the captured launches point at
        lib = _lib.load()

        # the local prior hands back its channel-major [B, 4, N] output; the update reads it in that layout (round 6: no
        # transposing ATen copy at the end of a step)
        n_pts, n_cls = getattr(model, "num_points", 0), getattr(model, "num_classes", 0)
        self.cm_out = (CHANNEL_MAJOR_EPS and hasattr(model, "geometry_source") and n_cls == 4
                       and int(np.prod(shape)) == n_pts * n_cls and not getattr(model, "mixed_prediction", False))

        def step():
            st = _lib.stream_ptr(dev)
            extra = {}
            if self.temb_table is not None:   # the step's time-embedding row, copied by the prologue kernel itself
                _lib.check(lib.lion_chain_begin_step_temb(_lib.ptr(self.table), self.capacity, _lib.ptr(self.counter),
                                                          _lib.ptr(self.t), num_samples, _lib.ptr(self.cur),
                                                          _lib.ptr(self.temb_table), self.temb_row.numel(),
                                                          _lib.ptr(self.temb_row), st), "chain_begin_step_temb")
                extra["temb"] = self.temb_row
            else:
                _lib.check(lib.lion_chain_begin_step(_lib.ptr(self.table), self.capacity, _lib.ptr(self.counter),
                                                     _lib.ptr(self.t), num_samples, _lib.ptr(self.cur), st),
                           "chain_begin_step")
            if self.cm_out:
                pred = model(x=self.x, t=self.t, condition_input=self.cond, clip_feat=self.clip, channel_major_out=True,
                             **extra)
                eps = pred.float().contiguous()
                assert tuple(eps.shape) == (num_samples, n_cls, n_pts)
                _lib.check(lib.lion_chain_update_noise_cm(mode, _lib.ptr(self.x), _lib.ptr(eps), num_samples, n_pts,
                                                          _lib.ptr(self.cur), _lib.ptr(self.seed), 0, _lib.ptr(self.x),
                                                          _lib.ptr(self.z), _lib.stream_ptr(dev)), "chain_update_noise_cm")
                return
            pred = model(x=self.x, t=self.t, condition_input=self.cond, clip_feat=self.clip, **extra)
            eps = pred.float().contiguous()
            _lib.check(lib.lion_chain_update_noise(mode, _lib.ptr(self.x), _lib.ptr(eps), self.x.numel(),
                                                   _lib.ptr(self.cur), _lib.ptr(self.seed), 0, _lib.ptr(self.x),
                                                   _lib.ptr(self.z), _lib.stream_ptr(dev)), "chain_update_noise")

        # Split-graph geometry overlap (geometry.SPLIT_GRAPH, models with a geometry_source()): the FPS / ball-query chain of
        # a step depends on the coordinates of x alone, is latency bound (0.7 ms on 32 CUs) and, on one stream, serial.
        # A parallel BRANCH inside one hipGraph replays slower than one stream on ROCm 7.2 (geometry.py), but separate
        # graphs on separate streams do overlap.  So a step becomes three single-branch graphs:
        #     geometry stream:  [geo: coordinates of x -> FPS / ball-query chain]            (waits for the previous step)
        #     main stream:      [A: step prologue, forward up to the first use of a geometry result]
        #                       wait(geo)  [B: the rest of the forward, update + noise]
        # ordered with two events per step.  Models without set abstraction never reach "first use": one graph, as before.
        from . import geometry
        src = getattr(model, "geometry_source", None)
        self.geo_graphs = self.graph_b = None
        self.geo_plan = None
        split = geometry.SPLIT_GRAPH and src is not None and not geometry.ENABLED
        if self.use_temb_table:
            with torch.no_grad():
                row = model.time_embedding(torch.ones(1, device=dev))
            self.temb_table = torch.zeros((self.capacity,) + tuple(row.shape[1:]), device=dev)
            self.temb_row = torch.zeros((1,) + tuple(row.shape[1:]), device=dev)
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        with _wcache.pinning(self.pinned):
            with torch.no_grad(), torch.cuda.stream(side):
                for _ in range(warmup):   # first calls pack weights, set kernel attributes, fill caches
                    step()
            main.wait_stream(side)
            if split:
                # NORMAL priority: on a high-priority stream (geometry._side_stream) the same graphs take 15.3 ms per
                # step instead of 7.4 -- the main stream's queue starves while the 0.7-ms FPS kernels run (measured, round 3)
                self.geo_stream = torch.cuda.Stream(device=dev)
                self.ev_step = torch.cuda.Event()

                def geo(boundary=None):
                    mods, coords = src(self.x)
                    return geometry.compute_chain(mods, coords, boundary=boundary)
                self.geo_stream.wait_stream(main)
                with torch.no_grad(), torch.cuda.stream(self.geo_stream):
                    geo()
                main.wait_stream(self.geo_stream)
                torch.cuda.synchronize(dev)
                # geometry graphs: [stage 0: the 2048 -> 1024 FPS, 0.55 of the chain's 0.76 ms, and its ball queries] and
                # [the later stages] -- the forward waits for the first one only where it needs it (SA-1's grouping); the
                # second is long done when SA-2 asks (two cuts of the main graph instead of one: 0.2 ms less exposed)
                self.geo_graphs = [torch.cuda.CUDAGraph()]
                gs = torch.cuda.Stream(device=dev)
                gs.wait_stream(main)

                def geo_cut(i):
                    if i == 1:
                        self.geo_graphs[-1].capture_end()
                        self.geo_graphs.append(torch.cuda.CUDAGraph())
                        self.geo_graphs[-1].capture_begin(pool=self.geo_graphs[0].pool())
                with torch.no_grad(), torch.cuda.stream(gs):
                    self.geo_graphs[0].capture_begin()
                    try:
                        self.geo_plan = geo(geo_cut)
                    finally:
                        self.geo_graphs[-1].capture_end()
                main.wait_stream(gs)
                torch.cuda.synchronize(dev)
                self.geo_stage_of_graph = [0, self.geo_plan["stages"] - 1][:len(self.geo_graphs)]  # last stage each graph holds
                self.ev_geo = [torch.cuda.Event() for _ in self.geo_graphs]
                graphs, waits = [torch.cuda.CUDAGraph()], []   # main graphs; waits[k]: geometry graphs awaited before graphs[k+1]

                def on_use(first, last):   # called inside the forward, on the capturing stream
                    need = [g for g, st_ in enumerate(self.geo_stage_of_graph)
                            if st_ >= first and (g == 0 or self.geo_stage_of_graph[g - 1] < last)]
                    need = [g for g in need if not any(g in w for w in waits)]
                    if not need:
                        return
                    graphs[-1].capture_end()
                    waits.append(need)
                    graphs.append(torch.cuda.CUDAGraph())
                    graphs[-1].capture_begin(pool=graphs[0].pool())
                cap = torch.cuda.Stream(device=dev)
                cap.wait_stream(main)
                with torch.no_grad(), torch.cuda.stream(cap), geometry.external(self.geo_plan, on_use):
                    graphs[0].capture_begin()
                    try:
                        step()
                    finally:
                        graphs[-1].capture_end()
                main.wait_stream(cap)
                self.graph = graphs[0]
                if waits:
                    self.graphs, self.waits = graphs, waits
                    self.graph_b = graphs[1]
                else:              # the forward never asked for a geometry result: one graph, no geometry stream
                    self.geo_graphs = self.geo_plan = None
            else:
                self.graph = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(self.graph):
                    step()
        self.pinned = list({id(v): v for v in self.pinned}.values())   # one reference per distinct object

    def replay(self):
        """one chain step on the current stream (+ the geometry stream in split mode)"""
        if self.graph_b is None:
            self.graph.replay()
            return
        main = torch.cuda.current_stream(self.x.device)
        self.ev_step.record(main)                 # x of this step is final (and last step's readers of the plan are done)
        self.geo_stream.wait_event(self.ev_step)
        with torch.cuda.stream(self.geo_stream):
            for g, ev in zip(self.geo_graphs, self.ev_geo):
                g.replay()
                ev.record(self.geo_stream)
        self.graphs[0].replay()                   # overlaps the geometry chain
        for k, need in enumerate(self.waits):
            for g in need:
                main.wait_event(self.ev_geo[g])
            self.graphs[k + 1].replay()

    def matches(self, condition_input, clip_feat):
        same = lambda buf, new: (buf is None) == (new is None) and (buf is None or buf.shape == new.shape)
        return same(self.cond, condition_input) and same(self.clip, clip_feat) \
            and self.fingerprint == _wcache.fingerprint(self.model) and self.policy == policy_key()

    @torch.no_grad()
    def run(self, x_init, table: np.ndarray, seed: int, condition_input=None, clip_feat=None, trajectory=None,
            trajectory_before_last=False, noise_trajectory=None, state_hook=None):
        """x_init: the chain's start; table [S, 8] float32 rows {t_model, a0..a5, 0}.  Returns the final latent
        (a fresh tensor).  `trajectory`: list that receives a copy of x after every step (before the last
        step's update if `trajectory_before_last`, as run_denoising_diffusion reports it).
        `state_hook(i, x)`: called before step i's replay with the chain's latent buffer, which it may overwrite IN PLACE
        with launches on the current stream (no host synchronisation): inpainting / known-region replacement, or a
        benchmark forcing the states a trained model would visit (bench.py's forced clouds)."""
        S = self.prepare(x_init, table, seed, condition_input, clip_feat)
        if RECORD is not None and self.z is not None and noise_trajectory is None:
            noise_trajectory = []
            RECORD.append((x_init.detach().clone(), noise_trajectory))
        for i in range(S):
            if trajectory is not None and trajectory_before_last and i == S - 1:
                trajectory.append(self.x.clone())
            if state_hook is not None:
                state_hook(i, self.x)
            self.replay()
            if noise_trajectory is not None and self.z is not None:
                noise_trajectory.append(self.z.clone())
            if trajectory is not None and not (trajectory_before_last and i == S - 1):
                trajectory.append(self.x.clone())
        return self.x.clone()

    @torch.no_grad()
    def prepare(self, x_init, table: np.ndarray, seed: int, condition_input=None, clip_feat=None) -> int:
        """everything a chain needs in device memory before its first replay (on the current stream): the start latent, the
        conditioning, the schedule table, the time-embedding rows, the step counter and the Philox key.  Returns S."""
        S = int(table.shape[0])
        assert 1 <= S <= self.capacity and table.shape[1] == 8
        self.x.copy_(x_init)
        if self.cond is not None:
            self.cond.copy_(condition_input)
        if self.clip is not None:
            self.clip.copy_(clip_feat)
        self.table[:S].copy_(torch.from_numpy(np.ascontiguousarray(table, dtype=np.float32)), non_blocking=False)
        if self.temb_table is not None:
            was_training = self.model.training
            self.model.eval()
            self.temb_table[:S].copy_(self.model.time_embedding(self.table[:S, 0].contiguous()))
            self.model.train(was_training)
        self.counter.zero_()
        words = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32).view(np.int32)
        self.seed.copy_(torch.from_numpy(words))
        return S


class ChainCache:
    """the captured chains of one DiffusionDiscretized: (model, mode, batch shape) -> GraphedChain; a handful of
    entries (each holds the private memory pool of one forward pass)."""

    def __init__(self, capacity=4):
        self._entries = {}
        self._order = []
        self._capacity = capacity

    def get(self, model, num_samples, shape, condition_input, clip_feat, device, mode, table_capacity):
        rec = RECORD is not None
        key = (id(model), mode, int(num_samples), tuple(shape), str(device), rec)
        hit = self._entries.get(key)
        if hit is not None and hit.model is model and hit.capacity >= table_capacity \
                and hit.matches(condition_input, clip_feat):
            self._order.remove(key)
            self._order.append(key)
            return hit
        if hit is not None:
            del self._entries[key]
            self._order.remove(key)
        chain = GraphedChain(model, num_samples, shape, condition_input, clip_feat, device, mode, table_capacity,
                             record_noise=rec)
        self._entries[key] = chain
        self._order.append(key)
        while len(self._order) > self._capacity:
            del self._entries[self._order.pop(0)]
        return chain

    def clear(self):
        self._entries.clear()
        self._order = []


def draw_seed() -> int:
    """64-bit Philox key for one chain, drawn from torch's default CPU generator: torch.manual_seed() keeps
    governing reproducibility, as it does for the reference's per-step randn."""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF


def graphable(model, x, enable_autocast, extra_kwargs) -> bool:
    """the chain graph covers the plain sampling configuration of every released model; anything else (mixed
    prediction, autocast, grid embeddings, CPU tensors) takes the eager loop."""
    return (x.is_cuda and not enable_autocast and not extra_kwargs
            and not getattr(model, "mixed_prediction", False) and not torch.is_grad_enabled())
