"""lion_amd.optim.Adam (csrc/optim.hip: the whole update in one launch) against torch.optim.Adam, the reference's optimizer
(utils/utils.py:115-121), and inside captured training steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    shapes = [(1,), (7,), (33, 5), (4097,), (64, 64, 3, 3, 3), (100001,), (2, 3, 4, 5), (256, 35, 1, 1)]
    return [torch.nn.Parameter(torch.randn(*s, device="cuda", generator=g) * 0.5) for s in shapes]


def _grads(params, step, seed, skip=()):
    g = torch.Generator(device="cuda").manual_seed(1000 * seed + step)
    out = []
    for i, p in enumerate(params):
        if i in skip:
            out.append(None)
            continue
        # odd gradients too: a view at a 4-byte offset of a larger buffer (not 16-byte aligned: the scalar path of the kernel)
        big = torch.randn(p.numel() + 3, device="cuda", generator=g)
        out.append(big[1:1 + p.numel()].view_as(p) if i % 2 else big[:p.numel()].view_as(p).clone())
    return out


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_matches_torch_adam(wd):
    from lion_amd.optim import Adam
    pa, pb = _params(3), _params(3)
    oa = Adam(pa, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, foreach=False, fused=False)
    for step in range(6):
        skip = (2, 5) if step in (1, 2) else ()        # parameters without a gradient are skipped and their step count falls behind
        for ps in (pa, pb):
            for p, g in zip(ps, _grads(ps, step, 7, skip)):
                p.grad = g
        oa.step()
        ob.step()
        if step == 3:                                   # a learning-rate change reaches the device
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 1e-3
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(pa, pb)):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == (4.0 if i in (2, 5) else 6.0)
        for name, x, y in (("param", a, b), ("exp_avg", sa["exp_avg"], sb["exp_avg"]), ("exp_avg_sq", sa["exp_avg_sq"], sb["exp_avg_sq"])):
            err = (x.detach() - y.detach()).abs().max().item()
            assert err <= 2e-6 * max(y.detach().abs().max().item(), 1e-6), (i, name, err)


def test_adam_state_dict_moves_between_the_two():
    from lion_amd.optim import Adam
    pa, pb = _params(5), _params(5)
    ob = torch.optim.Adam(pb, lr=1e-3, betas=(0.9, 0.99))
    for step in range(2):
        for p, g in zip(pb, _grads(pb, step, 9)):
            p.grad = g
        ob.step()
    oa = Adam(pa, lr=1e-3, betas=(0.9, 0.99))
    with torch.no_grad():
        for a, b in zip(pa, pb):
            a.copy_(b)
    import copy
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))  # (state_dict() hands out the live tensors) host-side step counts become device counters
    for ps, o in ((pa, oa), (pb, ob)):
        for p, g in zip(ps, _grads(ps, 2, 9)):
            p.grad = g
        o.step()
    for a, b in zip(pa, pb):
        assert (a.detach() - b.detach()).abs().max().item() <= 2e-6 * b.detach().abs().max().item()
    ob2 = torch.optim.Adam(_params(5), lr=1e-3, betas=(0.9, 0.99), capturable=True)
    ob2.load_state_dict(copy.deepcopy(oa.state_dict()))  # and back
    assert float(next(iter(ob2.state.values()))["step"]) == 3.0


def test_adam_rejects_what_it_does_not_implement():
    from lion_amd.optim import Adam
    p = torch.nn.Parameter(torch.randn(8, device="cuda", dtype=torch.float64))
    p.grad = torch.randn_like(p)
    with pytest.raises(RuntimeError, match="float32"):
        Adam([p]).step()


def test_adam_inside_captured_training_steps():
    """GraphedTrainStep with lion_amd.optim.Adam: whole-step graph, [forward + backward] / [optimizer] graphs and the eager loop
    leave bit-identical parameters (the pointer table is written inside the capture from pinned memory: a copy node of the graph)."""
    from lion_amd.dist import BucketedGradAverager
    from lion_amd.optim import Adam
    from lion_amd.training import GraphedTrainStep

    def run(mode):
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 4)).cuda()
        unused = torch.nn.Parameter(torch.ones(7, device="cuda"))
        params = list(net.parameters()) + [unused]
        opt = Adam(params, lr=1e-2, betas=(0.9, 0.99), weight_decay=1e-3)
        avg = BucketedGradAverager(params, bucket_bytes=2048)
        gen = torch.Generator(device="cuda").manual_seed(5)
        xs = [torch.randn(32, 16, device="cuda", generator=gen) for _ in range(8)]
        ys = [torch.randn(32, 4, device="cuda", generator=gen) for _ in range(8)]

        def fb(x, y):
            avg.zero_grad()
            loss = ((net(x) - y) ** 2).mean()
            loss.backward()
            return loss.detach(), None
        if mode == "reference":
            for i in range(3, 8):
                fb(xs[i], ys[i])
                avg.finish()
                opt.step()
            torch.cuda.synchronize()
            return None, [p.detach().clone() for p in net.parameters()]
        st = GraphedTrainStep(fb, {"x": xs[0].clone(), "y": ys[0].clone()}, params, opt, avg, mode=mode, warmup=3)
        for p_ in net.parameters():                     # construction put the consumed steps back
            assert float(opt.state[p_]["step"]) == 0.0 and not bool(opt.state[p_]["exp_avg"].any())
        for i in range(3, 8):
            st(x=xs[i], y=ys[i])
        torch.cuda.synchronize()
        assert unused.grad is None and torch.equal(unused.detach(), torch.ones(7, device="cuda")) and unused not in opt.state
        return st, [p.detach().clone() for p in net.parameters()]

    st_w, p_w = run("whole")
    st_s, p_s = run("split")
    _, p_r = run("reference")
    assert st_w.mode == "whole" and len(st_w._graphs) == 1, st_w.launch
    assert st_s.mode == "split" and len(st_s._graphs) == 2, st_s.launch
    for a, b, c in zip(p_w, p_s, p_r):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_adam_keeps_the_moving_average_of_the_weights_like_the_reference_wrapper():
    """ema_decay: state[p]['ema'] == what utils/ema.py:47-78 computes around torch.optim.Adam (started from the updated parameter at a
    parameter's first step, ema * decay + (1 - decay) * param after every later one; parameters without a gradient untouched);
    lion_amd.training.EMA around the optimizer hands its decay over and swaps the weights in."""
    from lion_amd.optim import Adam
    from lion_amd.training import EMA
    decay = 0.99
    pa, pb = _params(3), _params(3)
    oa = EMA(Adam(pa, lr=3e-3, betas=(0.9, 0.99)), decay)
    assert oa._folded and oa.optimizer.ema_decay == decay
    ob = torch.optim.Adam(pb, lr=3e-3, betas=(0.9, 0.99), foreach=False, fused=False)
    ema_ref = {}
    for step in range(5):
        skip = (1, 4) if step == 0 else ()                 # these two take their first step one step later
        for ps in (pa, pb):
            for p, g in zip(ps, _grads(ps, step, 11, skip)):
                p.grad = g
        oa.step()
        ob.step()
        for i, p in enumerate(pb):                          # the reference wrapper's arithmetic
            if p.grad is None:
                continue
            if i not in ema_ref:
                ema_ref[i] = p.detach().clone()
            ema_ref[i].mul_(decay).add_(p.detach(), alpha=1.0 - decay)
    torch.cuda.synchronize()
    for i, a in enumerate(pa):
        e = oa.state[a]["ema"]
        assert (e - ema_ref[i]).abs().max().item() <= 2e-6 * max(ema_ref[i].abs().max().item(), 1e-6), i
        assert (e - a.detach()).abs().max().item() > 0      # an average, not a copy
    before = [p.detach().clone() for p in pa]
    oa.swap_parameters_with_ema(store_params_in_ema=True)
    for i, a in enumerate(pa):
        assert (a.detach() - ema_ref[i]).abs().max().item() <= 2e-6 * max(ema_ref[i].abs().max().item(), 1e-6)
        assert torch.equal(oa.state[a]["ema"], before[i])
