"""-m gpu: the graphed sampling chain (lion_amd/chain.py, csrc/diffusion.hip): on-chip Philox noise against a numpy
restatement, the fused update against lion_ddim_update / lion_ddpm_update on the same noise, and graph replay
against the eager per-step sampler."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c, k):
    """numpy restatement (Salmon et al., SC'11); c: [n,4] uint32 counters, k: (k0, k1)."""
    c = c.astype(np.uint64)
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    for _ in range(10):
        p0 = np.uint64(M0) * c[:, 0]
        p1 = np.uint64(M1) * c[:, 2]
        c = np.stack([((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & np.uint64(0xFFFFFFFF), p1 & np.uint64(0xFFFFFFFF),
                      ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & np.uint64(0xFFFFFFFF), p0 & np.uint64(0xFFFFFFFF)], 1)
        k0 = (k0 + np.uint64(W0)) & np.uint64(0xFFFFFFFF)
        k1 = (k1 + np.uint64(W1)) & np.uint64(0xFFFFFFFF)
    return c.astype(np.uint32)


def normals(numel, step, stream_id, seed):
    quads = (numel + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    ctr = np.stack([q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.full(quads, step, np.uint64),
                    np.full(quads, stream_id, np.uint64)], 1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    u = (r.astype(np.float32) * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)).astype(np.float32)
    out = np.empty((quads, 4), np.float64)
    for h in range(2):
        rad = np.sqrt(-2.0 * np.log(u[:, 2 * h].astype(np.float64)))
        ang = 2.0 * np.pi * u[:, 2 * h + 1].astype(np.float64)
        out[:, 2 * h] = rad * np.cos(ang)
        out[:, 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:numel]


def test_philox_known_answer():
    """Random123 kat_vectors: philox4x32-10, counter 0 key 0 / all ones / pi digits."""
    assert [hex(v) for v in philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]] == \
        ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    ones = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    assert [hex(v) for v in philox4x32_10(ones, (0xFFFFFFFF, 0xFFFFFFFF))[0]] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    pi = np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32)
    assert [hex(v) for v in philox4x32_10(pi, (0xa4093822, 0x299f31d0))[0]] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def _call_update(mode, x, eps, row, step, seed, want_z=True):
    from lion_amd import _lib
    lib = _lib.load()
    cur = torch.tensor(list(row[:7]) + [0.0], device="cuda")
    cur[7] = torch.tensor([step], dtype=torch.int32).view(torch.float32)[0]
    words = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32).view(np.int32)
    sd = torch.from_numpy(words).cuda()
    out, z = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.lion_chain_update_noise(mode, _lib.ptr(x), _lib.ptr(eps), x.numel(), _lib.ptr(cur), _lib.ptr(sd),
                                           0, _lib.ptr(out), _lib.ptr(z) if want_z else None,
                                           _lib.stream_ptr(x.device)), "chain_update_noise")
    return out, z


@pytest.mark.parametrize("numel", [32 * 8192, 32 * 128, 1001])
def test_chain_update_noise(numel):
    from lion_amd.diffusion_ops import ddim_update, ddpm_update
    g = torch.Generator(device="cuda").manual_seed(numel)
    x = torch.randn(numel, device="cuda", generator=g)
    e = torch.randn(numel, device="cuda", generator=g)
    seed, step = 0x1234567890ABCDEF, 417
    s, c, sg = 1.0001, -0.0123, 0.0456
    out, z = _call_update(0, x, e, (418.0, s, c, sg, 0, 0, 0), step, seed)
    # the noise: bit-exact Philox integers -> Box-Muller; device logf / sincospif vs float64 libm
    ref = normals(numel, step, 0, seed)
    np.testing.assert_allclose(z.cpu().numpy().astype(np.float64), ref, rtol=0, atol=2e-6)
    if numel > 100000:
        zz = z.double()
        assert abs(zz.mean().item()) < 0.01 and abs(zz.var().item() - 1.0) < 0.01
        assert abs((zz ** 3).mean().item()) < 0.03 and abs((zz ** 4).mean().item() - 3.0) < 0.06
    # the update: same arithmetic as lion_ddim_update on that noise, bit for bit
    assert torch.equal(out, ddim_update(x, e, z, s, c, sg))
    # another step / seed -> another stream
    _, z2 = _call_update(0, x, e, (418.0, s, c, sg, 0, 0, 0), step + 1, seed)
    _, z3 = _call_update(0, x, e, (418.0, s, c, sg, 0, 0, 0), step, seed + 1)
    assert not torch.equal(z, z2) and not torch.equal(z, z3)
    # DDPM rows
    ko, ka, kb, sc, temp = 1.00005, 1e-4, 0.83, 0.01, 0.9
    out, z = _call_update(1, x, e, (5.0, ko, ka, kb, sc, temp, 0.0), 3, seed)
    assert torch.equal(out, ddpm_update(x, e, z, False, ko, ka, kb, sc, temp))
    out, _ = _call_update(1, x, e, (1.0, ko, ka, 1.0, sc, temp, 1.0), 3, seed)
    assert torch.equal(out, ddpm_update(x, e, None, True, ko, ka, 1.0, sc, temp))
    # in place (out aliases x), without z_out
    from lion_amd import _lib
    x2 = x.clone()
    cur = torch.tensor([1.0, s, c, sg, 0, 0, 0, 0], device="cuda")
    sd = torch.zeros(2, dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().lion_chain_update_noise(0, _lib.ptr(x2), _lib.ptr(e), numel, _lib.ptr(cur), _lib.ptr(sd), 0,
                                                   _lib.ptr(x2), None, _lib.stream_ptr(x.device)), "in place")
    o3, _ = _call_update(0, x, e, (1.0, s, c, sg, 0, 0, 0), 0, 0)
    assert torch.equal(x2, o3)


def test_begin_step_walks_the_table():
    from lion_amd import _lib
    lib = _lib.load()
    S, B = 5, 7
    table = torch.arange(S * 8, dtype=torch.float32, device="cuda").view(S, 8)
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    t, cur = torch.zeros(B, device="cuda"), torch.zeros(8, device="cuda")
    for i in range(S + 2):                        # two calls beyond the end: clamped to the last row
        _lib.check(lib.lion_chain_begin_step(_lib.ptr(table), S, _lib.ptr(counter), _lib.ptr(t), B, _lib.ptr(cur),
                                             _lib.stream_ptr(t.device)), "begin_step")
        j = min(i, S - 1)
        assert torch.equal(t, torch.full((B,), float(j * 8), device="cuda"))
        assert torch.equal(cur[:7], table[j, :7])
        assert cur[7:].view(torch.int32).item() == j
        assert counter.item() == j + 1


@pytest.fixture(scope="module")
def small_lion():
    from lion_amd.config import released_prior_cfg
    from lion_amd.models.lion import LION
    torch.manual_seed(3)
    lion = LION(released_prior_cfg())
    lion.priors.eval()
    lion.vae.eval()
    return lion


def test_graphed_chain_equals_eager_steps(small_lion):
    """Every replay of the captured step == the eager step on the same state and the same noise: x_{i+1} from the
    graph vs lion_ddim_update(x_i, model(x_i, t_i), z_i) evaluated eagerly."""
    from lion_amd import chain, diffusion_ops
    lion, d = small_lion, small_lion.diffusion
    B, S = 2, 5
    sh = lion.vae.latent_shape()
    with torch.no_grad():
        style = lion.vae.global2style(torch.randn([B] + sh[0], device="cuda"))
        for prior, shape, cond in ((lion.priors[0], sh[0], None), (lion.priors[1], sh[1], style)):
            steps = d.ddim_schedule(1000, S, 'uniform')
            table = np.zeros((S, 8), np.float32)
            for i, t in enumerate(steps):
                table[i, :4] = (t + 1,) + d.ddim_coefficients(t, None if i == S - 1 else steps[i + 1], 1.0)
            ch = chain.GraphedChain(prior, B, shape, cond, None, "cuda", chain.DDIM, 16, record_noise=True)
            x0 = torch.randn([B] + shape, device="cuda")
            xs, zs = [], []
            xf = ch.run(x0, table, 99, cond, None, trajectory=xs, noise_trajectory=zs)
            assert len(xs) == S and torch.equal(xf, xs[-1])
            x = x0
            for i in range(S):
                tt = torch.full((B,), float(table[i, 0]), device="cuda")
                eps = prior(x=x, t=tt, condition_input=cond, clip_feat=None).float().contiguous()
                want = diffusion_ops.ddim_update(x, eps, zs[i], *[float(v) for v in table[i, 1:4]])
                assert torch.equal(xs[i], want), (i, (xs[i] - want).abs().max().item() / want.abs().max().item())
                x = xs[i]                            # follow the graph's trajectory: per-step comparison


def test_product_sampler_graph_equals_eager_when_deterministic(small_lion):
    """run_ddim(graph=True) vs run_ddim(graph=False) with kappa = 0 (sigma = 0: no noise enters) from the same start;
    and generate_samples_vada_2prior runs end to end on the graphed path; a second call reuses the capture."""
    from lion_amd.sampling import generate_samples_vada_2prior
    lion, d = small_lion, small_lion.diffusion
    B = 2
    sh = lion.vae.latent_shape()
    x0 = torch.randn([B] + sh[1], device="cuda")
    style = lion.vae.global2style(torch.randn([B] + sh[0], device="cuda"))
    a, tr = d.run_ddim(lion.priors[1], B, sh[1], ddim_step=4, kappa=0.0, condition_input=style, x_noisy=x0,
                       is_image=False, graph=True)
    b, _ = d.run_ddim(lion.priors[1], B, sh[1], ddim_step=4, kappa=0.0, condition_input=style, x_noisy=x0,
                      is_image=False, graph=False)
    assert len(tr) == 4 and torch.equal(tr[-1], a)
    assert torch.equal(a, b), (a - b).abs().max().item() / b.abs().max().item()
    n_before = len(d._chains._entries)
    torch.manual_seed(11)
    p1, _ = generate_samples_vada_2prior(sh, lion.priors, d, lion.vae, B, ddim_step=3)
    n_mid = len(d._chains._entries)
    torch.manual_seed(11)
    p2, _ = generate_samples_vada_2prior(sh, lion.priors, d, lion.vae, B, ddim_step=3)
    assert tuple(p1.shape) == (B, 2048, 3) and torch.isfinite(p1).all()
    assert len(d._chains._entries) == n_mid >= n_before     # second call: no new capture
    # same torch seed -> same start and same Philox key -> the same cloud, bit for bit (no kernel of the step leaves its
    # summation order to the scheduler: tools/determinism_probe.py)
    assert torch.equal(p1, p2)


def test_chain_is_recaptured_when_weights_change(small_lion):
    lion, d = small_lion, small_lion.diffusion
    B = 2
    sh = lion.vae.latent_shape()
    x0 = torch.randn([B] + sh[0], device="cuda")
    a, _ = d.run_ddim(lion.priors[0], B, sh[0], ddim_step=3, kappa=0.0, x_noisy=x0, is_image=False)
    with torch.no_grad():
        for p in lion.priors[0].parameters():
            p.mul_(1.01)
    b, _ = d.run_ddim(lion.priors[0], B, sh[0], ddim_step=3, kappa=0.0, x_noisy=x0, is_image=False)
    c, _ = d.run_ddim(lion.priors[0], B, sh[0], ddim_step=3, kappa=0.0, x_noisy=x0, is_image=False, graph=False)
    assert not torch.equal(a, b)
    assert (b - c).abs().max().item() <= 1e-5 * c.abs().max().item()


def test_lion_sample_ancestral_chain(small_lion):
    """LION.sample (demo.py path): the ancestral chain through the graphed DDPM step; the scheduler shim's variance is
    the posterior one ('fixedlarge' is not a diffusers name, see models/lion.py)."""
    lion = small_lion
    d = lion.diffusion
    steps = d._diffusion_steps
    try:
        d._diffusion_steps = 6                      # keep the test short: a 6-step chain on the same tables
        out = lion.sample(num_samples=2)
    finally:
        d._diffusion_steps = steps
    assert tuple(out['points'].shape) == (2, 2048, 3) and torch.isfinite(out['points']).all()
    sch = lion.scheduler
    t = 500
    ab = d._h_alpha_bars.double()
    post = d._h_betas.double()[t] * (1 - ab[t - 1]) / (1 - ab[t])
    assert abs(sch._noise_scale(t) - float(post.sqrt())) < 1e-6
    sch.variance_type = 'fixed_large'
    assert abs(sch._noise_scale(t) - float(d._h_betas[t].sqrt())) < 1e-7
    sch.variance_type = 'fixedlarge'


def test_split_graph_chain_equals_the_single_graph_chain(small_lion):
    """The local prior's chain cut into [geometry stage 0 | later stages] on a second stream and [A | B | C] on the main one
    (lion_amd/chain.py, geometry.SPLIT_GRAPH) against the same chain as ONE graph on one stream: the same kernels on the same
    data in a different launch structure -- every step of the trajectory bit for bit, at B = 2 and at B = 32."""
    from lion_amd import chain, geometry
    lion, d = small_lion, small_lion.diffusion
    S = 6
    sh = lion.vae.latent_shape()
    saved = geometry.SPLIT_GRAPH
    try:
        for B in (2, 32):
            with torch.no_grad():
                style = lion.vae.global2style(torch.randn([B] + sh[0], device="cuda"))
            steps = d.ddim_schedule(1000, S, 'uniform')
            table = np.zeros((S, 8), np.float32)
            for i, t in enumerate(steps):
                table[i, :4] = (t + 1,) + d.ddim_coefficients(t, None if i == S - 1 else steps[i + 1], 1.0)
            x0 = torch.randn([B] + sh[1], device="cuda")
            trajs = []
            for split in (True, False):
                geometry.SPLIT_GRAPH = split
                ch = chain.GraphedChain(lion.priors[1], B, sh[1], style, None, "cuda", chain.DDIM, 16)
                assert (ch.graph_b is not None) == split, "split mode must cut the step where a geometry result is first used"
                if split:
                    assert len(ch.geo_graphs) == 2 and len(ch.graphs) == 3
                xs = []
                for _ in range(2):                    # the chain twice: replays 1..12 of one capture
                    xs = []
                    ch.run(x0, table, 1234, style, None, trajectory=xs)
                trajs.append(xs)
                del ch
            for i, (a, b) in enumerate(zip(*trajs)):
                assert torch.equal(a, b), (B, i, (a - b).abs().max().item())
    finally:
        geometry.SPLIT_GRAPH = saved



def test_state_hook_graph_equals_eager(small_lion):
    """run_ddim(state_hook=...) (round 6; bench.py's forced clouds, known-region replacement): the hook overwrites the chain's
    latent before every model evaluation, in the graphed chain and in the eager loop alike.  With kappa = 0 (no noise enters) the
    two paths give the same final latent bit for bit, the hook is called once per step with ascending step indices, and the result
    differs from the un-hooked chain."""
    lion, d = small_lion, small_lion.diffusion
    B, S = 2, 4
    sh = lion.vae.latent_shape()
    style = lion.vae.global2style(torch.randn([B] + sh[0], device="cuda"))
    x0 = torch.randn([B] + sh[1], device="cuda")
    forced = [torch.randn([B] + sh[1], device="cuda") for _ in range(S)]
    outs = {}
    for graph in (True, False):
        seen = []

        def hook(i, x):
            seen.append(i)
            x.copy_(forced[i])
        outs[graph], _ = d.run_ddim(lion.priors[1], B, sh[1], ddim_step=S, kappa=0.0, condition_input=style, x_noisy=x0.clone(),
                                    is_image=False, graph=graph, state_hook=hook, keep_trajectory=False)
        assert seen == list(range(S))
    plain, _ = d.run_ddim(lion.priors[1], B, sh[1], ddim_step=S, kappa=0.0, condition_input=style, x_noisy=x0.clone(),
                          is_image=False, graph=True, keep_trajectory=False)
    assert torch.equal(outs[True], outs[False])
    assert not torch.equal(outs[True], plain)


def test_sampler_state_hook_names_the_prior(small_lion):
    from lion_amd.sampling import generate_samples_vada_2prior
    lion, d = small_lion, small_lion.diffusion
    calls = []
    generate_samples_vada_2prior(lion.vae.latent_shape(), lion.priors, d, lion.vae, 2, ddim_step=3,
                                 state_hook=lambda prior, i, x: calls.append((prior, i, tuple(x.shape))))
    assert [c[:2] for c in calls] == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]
    assert calls[0][2][0] == 2 and calls[3][2][1] == 2048 * 4
