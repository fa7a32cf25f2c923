"""-m gpu: the split-operand 3x3x3 convolution (csrc/conv3d_split.hip: fp16 x 2 pieces on the 16-bit MFMA pipe, power-of-
two block scaling) held to the SAME bounds as the exact-fp32 MFMA kernel -- max error < 5e-6 of the output's maximum
against a float64 convolution of the same fp32 operands -- in every mode (plain, AdaGN+Swish prologue + GroupNorm sums,
sparse, constant + delta), and on adversarial dynamic ranges where a naive fp16 cut would overflow, flush or lose the
residual: values beyond 65504, 1e-30, nine decades inside one tensor, samples of very different magnitude in one batch,
tiny / huge weights.  The fp32 kernel is the fallback (Cin % 16 != 0) and the reference arm of these tests."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BOUND = 5e-6   # the fp32 kernel's bound in test_hip_parity_gpu.py::test_conv3d_mfma_matches_fp64_reference


def ref64(x, conv, pro=None):
    xin = x.double()
    if pro is not None:
        A, Bs = pro
        xin = F.silu(xin * A.double().view(*A.shape, 1, 1, 1) + Bs.double().view(*Bs.shape, 1, 1, 1))
    bias = conv.bias.double() if conv.bias is not None else None
    return F.conv3d(xin, conv.weight.double(), bias, padding=1)


def rel_err(got, ref, per_sample=False):
    d = (got.double() - ref).abs()
    if per_sample:
        return (d.flatten(1).amax(1) / ref.abs().flatten(1).amax(1)).max().item()
    return d.max().item() / ref.abs().max().item()


def make(cin, cout, r, B=2, seed=0, bias=True):
    torch.manual_seed(seed + cin + cout + r)
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1, bias=bias).cuda()
    x = torch.randn(B, cin, r, r, r, device="cuda")
    return conv, x


@pytest.mark.parametrize("cin,cout,r", [(64, 64, 32), (32, 32, 32), (128, 64, 16), (16, 32, 16), (192, 128, 8),
                                        (256, 128, 8)])
def test_split_matches_fp64_and_fp32_kernel(cin, cout, r):
    from lion_amd.conv_ops import conv3d_k3
    conv, x = make(cin, cout, r)
    with torch.no_grad():
        ref = ref64(x, conv)
        got = conv3d_k3(x, conv.weight, conv.bias, split=True)
        f32 = conv3d_k3(x, conv.weight, conv.bias, split=False)
    e_split, e_f32 = rel_err(got, ref), rel_err(f32, ref)
    assert e_split < BOUND, (e_split, e_f32)
    assert e_split < 2.0 * e_f32 + 1e-7, (e_split, e_f32)     # not a precision reduction: same class as the fp32 chain
    scale = ref.abs().max().item()
    for sl in ((..., 0, slice(None), slice(None)), (..., slice(None), slice(None), r - 1), (..., 0, 0, 0),
               (..., r - 1, r - 1, r - 1)):                    # zero padding: faces and corners
        assert torch.allclose(got[sl].double(), ref[sl], rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("cin,cout,r", [(64, 64, 32), (128, 128, 16), (128, 128, 8)])
def test_split_prologue_and_groupnorm_sums(cin, cout, r):
    """swish(x * A + Bs) applied while staging + per-tile channel sums in the epilogue (conv3d_fused's PRO / STATS)."""
    from lion_amd import _lib, fused_ops as fo
    conv, x = make(cin, cout, r, B=3, seed=1)
    A = torch.rand(3, cin, device="cuda") + 0.5
    Bs = torch.randn(3, cin, device="cuda") * 0.5
    with torch.no_grad():
        ref = ref64(x, conv, (A, Bs))
        y, st = fo.conv3d_fused(x, conv, (A, Bs), True, None, split=True)
        y32, st32 = fo.conv3d_fused(x, conv, (A, Bs), True, None, split=False)
    # the prologue's swish uses v_exp / v_rcp (as in the fp32 kernel): the bound of the fused fp32 path applies
    assert rel_err(y, ref) < 1e-5 and rel_err(y, y32.double()) < 1e-5
    assert st.shape[2] == _lib.load().lion_conv3d_split_stat_tiles(r, cout)
    sums = st.double().sum(2)
    n = r ** 3
    scale = ref.abs().max().item()
    assert torch.allclose(sums[..., 0], ref.flatten(2).sum(-1), rtol=1e-4, atol=1e-5 * scale * n ** 0.5)
    assert torch.allclose(sums[..., 1], ref.flatten(2).square().sum(-1), rtol=1e-4)


def test_split_scale_invariance_is_exact():
    """Block scaling by powers of two: conv(x * 2^k) == conv(x) * 2^k and conv_{w * 2^k}(x) == conv_w(x) * 2^k bit for
    bit (no bias), for magnitudes from 1e-24 to 1e+24 -- there is no range in which the cut clamps, flushes or changes
    its rounding."""
    from lion_amd.conv_ops import conv3d_k3
    conv, x = make(64, 32, 16, bias=False)
    with torch.no_grad():
        base = conv3d_k3(x, conv.weight, None, split=True)
        for k in (-80, -40, -17, 9, 16, 40, 80):
            f = 2.0 ** k
            assert torch.equal(conv3d_k3(x * f, conv.weight, None, split=True), base * f), k
            w2 = (conv.weight * f).contiguous()
            assert torch.equal(conv3d_k3(x, w2, None, split=True), base * f), k
        assert torch.isfinite(base).all()


def _log_uniform(shape, lo, hi, gen):
    mag = torch.exp(torch.empty(shape, device="cuda").uniform_(float(torch.tensor(lo).log()), float(torch.tensor(hi).log()),
                                                             generator=gen))
    sign = torch.where(torch.rand(shape, device="cuda", generator=gen) < 0.5, -1.0, 1.0)
    return mag * sign


@pytest.mark.parametrize("case", ["nine-decades", "beyond-fp16-max", "tiny", "per-sample-scales", "huge-one-chunk",
                                  "tiny-weights", "huge-weights", "residual-bits"])
def test_split_adversarial_dynamic_range(case):
    from lion_amd.conv_ops import conv3d_k3
    gen = torch.Generator(device="cuda").manual_seed(11)
    cin, cout, r, B = 64, 64, 16, 4
    conv, x = make(cin, cout, r, B=B, seed=5)
    per_sample = False
    with torch.no_grad():
        if case == "nine-decades":            # |x| log-uniform over 1e-6 .. 1e4 inside every tile
            x = _log_uniform(x.shape, 1e-6, 1e4, gen)
        elif case == "beyond-fp16-max":       # every activation far above 65504
            x = x * 3.0e6
        elif case == "tiny":                  # fp16 would flush all of it
            x = x * 1e-30
        elif case == "per-sample-scales":     # the per-tile scale keeps each sample's own precision
            x = x * torch.tensor([1e-6, 1.0, 1e4, 3e7], device="cuda").view(B, 1, 1, 1, 1)
            per_sample = True
        elif case == "huge-one-chunk":        # one 16-channel chunk dominates: the running scale must follow it
            x[:, 32:48] *= 1e4
            x[:, :16] *= 1e-3
        elif case == "tiny-weights":
            conv.weight.mul_(1e-9)
            conv.bias.mul_(1e-9)
        elif case == "huge-weights":
            conv.weight.mul_(1e6)
        elif case == "residual-bits":         # values whose low piece alone carries the information: 1 + k * 2^-22
            k = torch.randint(-2048, 2048, x.shape, device="cuda", generator=gen).float()
            x = (1.0 + k * 2.0 ** -22) * torch.where(torch.rand(x.shape, device="cuda", generator=gen) < 0.5, -1.0, 1.0)
        ref = ref64(x, conv)
        got = conv3d_k3(x.contiguous(), conv.weight, conv.bias, split=True)
        f32 = conv3d_k3(x.contiguous(), conv.weight, conv.bias, split=False)
    assert torch.isfinite(got).all()
    e, e32 = rel_err(got, ref, per_sample), rel_err(f32, ref, per_sample)
    assert e < BOUND, (case, e, e32)


def test_split_nonfinite_inputs_propagate():
    """an inf / nan activation poisons the outputs it reaches (as in fp32 arithmetic) and nothing else."""
    from lion_amd.conv_ops import conv3d_k3
    conv, x = make(32, 32, 16, B=2)
    x[0, 3, 2, 2, 2] = float("inf")
    x[1, 7, 12, 12, 12] = float("nan")
    xc = x.clone()
    xc[0, 3, 2, 2, 2] = 0.0
    xc[1, 7, 12, 12, 12] = 0.0
    with torch.no_grad():
        got = conv3d_k3(x, conv.weight, conv.bias, split=True)
        clean = ref64(xc, conv)
    assert not torch.isfinite(got[0, :, 2, 2, 2]).any() and not torch.isfinite(got[1, :, 12, 12, 12]).any()
    # everything outside the 3x3x3 neighbourhoods -- the rest of the same workgroup tiles included -- is as accurate
    # as without the poison (the non-finite values do not enter the tile's scale)
    mask = torch.ones_like(got, dtype=torch.bool)
    mask[0, :, 1:4, 1:4, 1:4] = False
    mask[1, :, 11:14, 11:14, 11:14] = False
    assert torch.isfinite(got[mask]).all()
    assert (got.double() - clean)[mask].abs().max().item() / clean.abs().max().item() < BOUND


def test_split_inside_the_denoiser_matches_fp32_convs(monkeypatch):
    """PVCNN2Prior forward with every eligible convolution on the split kernel vs the same forward on the fp32
    kernel (sparse / delta modes active): the outputs agree to the fused path's own tolerance."""
    from lion_amd import conv_ops
    from lion_amd.config import released_prior_cfg
    from lion_amd.models.lion import LION
    torch.manual_seed(5)
    lion = LION(released_prior_cfg())
    prior = lion.priors[1].eval()
    sh = lion.vae.latent_shape()
    b = 4
    x = torch.randn([b] + sh[1], device="cuda")
    style = lion.vae.global2style(torch.randn([b] + sh[0], device="cuda"))
    t = torch.full((b,), 321.0, device="cuda")
    outs = {}
    for mode in (False, True):
        monkeypatch.setattr(conv_ops, "SPLIT", mode)
        with torch.no_grad():
            outs[mode] = prior(x=x, t=t, condition_input=style, clip_feat=None).float()
    err = (outs[True] - outs[False]).abs().max().item() / outs[False].abs().max().item()
    assert err < 1e-4, err
