"""-m gpu: the split-operand 1x1 convolution (csrc/pwconv_split.hip: fp16 x 2 pieces on the 16-bit MFMA pipe, power-of-two
block scaling per weight tensor and per (column, 16-channel chunk)) held to the SAME bound as the fp32-MFMA kernels of
pwconv.hip -- max error < 1e-5 of the output's maximum against a float64 evaluation (test_hip_parity_gpu.py) -- with the
AdaGN+Swish prologue, the GroupNorm tile sums, odd Cin / Cout, ragged L, and on adversarial dynamic ranges."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BOUND = 1e-5


def _ref64(x, conv, pro=None):
    xin = x.double()
    if pro is not None:
        A, Bs = pro
        t = xin * A.double()[:, :, None] + Bs.double()[:, :, None]
        xin = t * torch.sigmoid(t)
    out = torch.einsum("oc,bcl->bol", conv.weight.double()[:, :, 0], xin)
    return out + conv.bias.double()[None, :, None] if conv.bias is not None else out


def _err(got, ref, per_column=False):
    d = (got.double() - ref).abs()
    if per_column:   # every column against its own maximum: the block scaling is per column
        return (d.amax(1) / ref.abs().amax(1).clamp_min(1e-300)).max().item()
    return d.max().item() / ref.abs().max().item()


@pytest.mark.parametrize("cin,cout,L,pro", [(35, 32, 4096, False), (32, 64, 4096, True), (67, 128, 1000, True),
                                            (131, 128, 333, False), (64, 256, 2048, True), (35, 32, 9000, False),
                                            (32, 64, 20000, True), (192, 128, 2048, True), (320, 256, 300, False),
                                            (259, 96, 77, True), (16, 4, 130, False), (64, 384, 1, False)])
def test_pwconv_split_matches_fp64_and_fp32_kernel(cin, cout, L, pro):
    from lion_amd import _lib
    from lion_amd import fused_ops as fo
    torch.manual_seed(cin + L)
    B = 3
    conv = torch.nn.Conv1d(cin, cout, 1).cuda()
    x = torch.randn(B, cin, L, device="cuda")
    A = torch.randn(B, cin, device="cuda") * 0.5 + 1.0
    Bs = torch.randn(B, cin, device="cuda") * 0.3
    p = (A, Bs) if pro else None
    with torch.no_grad():
        ref = _ref64(x, conv, p)
        y, st = fo.pwconv_fused(x, conv, p, split=True)
        y32, _ = fo.pwconv_fused(x, conv, p, split=False)
        y2, st2 = fo.pwconv_fused(x, conv, p, want_stats=False, split=True)
    assert st2 is None and torch.equal(y, y2) and tuple(y.shape) == (B, cout, L)
    e, e32 = _err(y, ref), _err(y32, ref)
    assert e < BOUND, (e, e32)
    assert e < 2.0 * e32 + 2e-7, (e, e32)          # same class as the fp32 MFMA chain, not a precision reduction
    scale = ref.abs().max().item()
    sums = st.double().sum(2)
    assert torch.allclose(sums[..., 0], ref.sum(-1), rtol=1e-4, atol=1e-4 * scale * max(L, 1) ** 0.5)
    assert torch.allclose(sums[..., 1], ref.square().sum(-1), rtol=1e-4, atol=1e-6)
    assert _lib.load().lion_pwconv_split_stat_tiles(cout, cin, L) == st.shape[2]


def test_pwconv_split_2d_activation_and_no_bias():
    """[B, C, M, U] (grouped neighbourhoods) and bias=None go through the same entry."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(35, 64, 1, bias=False).cuda()
    x = torch.randn(2, 35, 257, 32, device="cuda")
    with torch.no_grad():
        y, st = fo.pwconv_fused(x, conv, None, split=True)
        ref = torch.einsum("oc,bcmu->bomu", conv.weight.double()[:, :, 0, 0], x.double())
    assert tuple(y.shape) == (2, 64, 257, 32) and _err(y.flatten(2), ref.flatten(2)) < BOUND


def test_pwconv_split_scale_invariance_is_exact():
    """power-of-two block scaling: f(x * 2^k) == f(x) * 2^k and f_{w * 2^k}(x) == f_w(x) * 2^k bit for bit (no bias)."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(0)
    conv = torch.nn.Conv1d(96, 64, 1, bias=False).cuda()
    x = torch.randn(2, 96, 3000, device="cuda")
    with torch.no_grad():
        base = fo.pwconv_fused(x, conv, None, want_stats=False, split=True)[0]
        for k in (-80, -40, -17, 9, 16, 40, 80):
            f = 2.0 ** k
            assert torch.equal(fo.pwconv_fused(x * f, conv, None, want_stats=False, split=True)[0], base * f), k
            conv2 = torch.nn.Conv1d(96, 64, 1, bias=False).cuda()
            conv2.weight.copy_(conv.weight * f)
            assert torch.equal(fo.pwconv_fused(x, conv2, None, want_stats=False, split=True)[0], base * f), k
        assert torch.isfinite(base).all()


def _log_uniform(shape, lo, hi, gen):
    import math
    mag = torch.exp(torch.empty(shape, device="cuda").uniform_(math.log(lo), math.log(hi), generator=gen))
    sign = torch.where(torch.rand(shape, device="cuda", generator=gen) < 0.5, -1.0, 1.0)
    return mag * sign


@pytest.mark.parametrize("case", ["nine-decades", "beyond-fp16-max", "tiny", "per-column-scales", "huge-one-chunk",
                                  "tiny-weights", "huge-weights", "residual-bits"])
def test_pwconv_split_adversarial_dynamic_range(case):
    from lion_amd import fused_ops as fo
    gen = torch.Generator(device="cuda").manual_seed(11)
    torch.manual_seed(5)
    cin, cout, L, B = 128, 128, 2048, 3
    conv = torch.nn.Conv1d(cin, cout, 1).cuda()
    x = torch.randn(B, cin, L, device="cuda")
    per_column = False
    with torch.no_grad():
        if case == "nine-decades":
            x = _log_uniform(x.shape, 1e-6, 1e4, gen)
        elif case == "beyond-fp16-max":
            x = x * 3.0e6
        elif case == "tiny":
            x = x * 1e-30
            conv.bias.zero_()
        elif case == "per-column-scales":     # neighbouring columns 13 decades apart keep their own precision
            x = x * _log_uniform((1, 1, L), 1e-6, 1e7, gen).abs()
            conv.bias.zero_()
            per_column = True
        elif case == "huge-one-chunk":        # one 16-channel chunk dominates: the running scale must follow it
            x[:, 32:48] *= 1e4
            x[:, :16] *= 1e-3
        elif case == "tiny-weights":
            conv.weight.mul_(1e-9)
            conv.bias.mul_(1e-9)
        elif case == "huge-weights":
            conv.weight.mul_(1e6)
        elif case == "residual-bits":
            k = torch.randint(-2048, 2048, x.shape, device="cuda", generator=gen).float()
            x = (1.0 + k * 2.0 ** -22) * torch.where(torch.rand(x.shape, device="cuda", generator=gen) < 0.5, -1.0, 1.0)
        ref = _ref64(x, conv)
        got = fo.pwconv_fused(x.contiguous(), conv, None, want_stats=False, split=True)[0]
        f32 = fo.pwconv_fused(x.contiguous(), conv, None, want_stats=False, split=False)[0]
    assert torch.isfinite(got).all()
    e, e32 = _err(got, ref, per_column), _err(f32, ref, per_column)
    assert e < BOUND, (case, e, e32)


def test_pwconv_split_nonfinite_inputs_propagate():
    """an inf / nan activation poisons its own column (as in fp32 arithmetic) and nothing else."""
    from lion_amd import fused_ops as fo
    torch.manual_seed(2)
    conv = torch.nn.Conv1d(64, 64, 1).cuda()
    x = torch.randn(2, 64, 1500, device="cuda")
    x[0, 3, 17] = float("inf")
    x[1, 40, 900] = float("nan")
    xc = x.clone()
    xc[0, 3, 17] = 0.0
    xc[1, 40, 900] = 0.0
    with torch.no_grad():
        got = fo.pwconv_fused(x, conv, None, want_stats=False, split=True)[0]
        clean = _ref64(xc, conv)
    assert not torch.isfinite(got[0, :, 17]).any() and not torch.isfinite(got[1, :, 900]).any()
    mask = torch.ones_like(got, dtype=torch.bool)
    mask[0, :, 17] = False
    mask[1, :, 900] = False
    assert torch.isfinite(got[mask]).all()
    assert (got.double() - clean)[mask].abs().max().item() / clean.abs().max().item() < BOUND


def test_pwconv_split_follows_weight_updates():
    """the packed pieces are cached per weight version: an in-place update must be seen"""
    from lion_amd import fused_ops as fo
    torch.manual_seed(4)
    conv = torch.nn.Conv1d(32, 32, 1).cuda()
    x = torch.randn(2, 32, 4096, device="cuda")
    with torch.no_grad():
        a = fo.pwconv_fused(x, conv, None, want_stats=False, split=True)[0]
        conv.weight.mul_(2.0)
        conv.bias.mul_(2.0)
        b = fo.pwconv_fused(x, conv, None, want_stats=False, split=True)[0]
    assert torch.equal(b, a * 2.0)
