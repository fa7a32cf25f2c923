"""Numerical claim behind DESIGN.md section 7 item 0 (the split-operand convolution measured in tools/exp/): cutting
fp32 operands into an fp16 head and a 2^11-rescaled fp16 residual, multiplying the pieces with fp32 accumulation
(main = ah*bh, corr = ah*bl + al*bh, result = main + corr/2048) is NOT less accurate than the native fp32 MFMA chain --
the dropped al*bl term is 2^-22 relative while the fp32 chain rounds its accumulator 8x more often (K = 2 per MFMA
instead of 16).  Model: products exact inside one MFMA, one fp32 rounding of the accumulator per MFMA."""
import numpy as np


def _f16(x):
    return x.astype(np.float16).astype(np.float32)


def _chain(pairs, k_per_mfma, K):
    acc = np.zeros((pairs[0][0].shape[0], pairs[0][1].shape[1]), np.float32)
    for k0 in range(0, K, k_per_mfma):
        part = sum(a[:, k0:k0 + k_per_mfma].astype(np.float64) @ b[k0:k0 + k_per_mfma].astype(np.float64)
                   for a, b in pairs)
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def _split(t):
    hi = _f16(t)
    return hi, _f16((t - hi) * 2048.0)


def _rms_err(y, truth):
    return float(np.sqrt((((y.astype(np.float64) - truth) / np.sqrt((truth ** 2).mean())) ** 2).mean()))


def test_fp16_pair_split_is_at_least_as_accurate_as_the_fp32_mfma_chain():
    rs = np.random.RandomState(0)
    M, N, K = 96, 32, 27 * 64
    x = rs.randn(M, K).astype(np.float32)
    a = (x / (1 + np.exp(-x))).astype(np.float32)  # Swish outputs, 30 % empty voxels
    a[rs.rand(M, K) < 0.3] = 0
    w = (rs.randn(K, N) / np.sqrt(K)).astype(np.float32)
    truth = a.astype(np.float64) @ w.astype(np.float64)
    native = _rms_err(_chain([(a, w)], 2, K), truth)
    (ah, al), (wh, wl) = _split(a), _split(w)
    main, corr = _chain([(ah, wh)], 16, K), _chain([(ah, wl), (al, wh)], 16, K)
    split = _rms_err((main.astype(np.float64) + corr.astype(np.float64) / 2048.0).astype(np.float32), truth)
    head_only = _rms_err(main, truth)
    assert split <= native and split < 4e-7, (split, native)
    assert head_only > 1e-4  # the residual pieces are what buys the accuracy: fp16 alone is 3 decimal digits
    # pieces reconstruct the operand to 2^-22 relative wherever the head is a normal fp16 number
    big = np.abs(a) > 2.0 ** -14
    assert np.abs((ah + al / 2048.0 - a)[big] / a[big]).max() < 2.0 ** -21


def test_range_limits_of_the_unscaled_split_are_the_documented_ones():
    """|x| > 65504 overflows the fp16 head (hence the clamp / per-tile scale in the draft kernel); tiny values lose
    relative but not absolute accuracy (the rescaled residual still resolves 2^-36)."""
    with np.errstate(over="ignore", invalid="ignore"):
        hi, lo = _split(np.array([7.0e4, 1.0e-6, 3.0e-9], np.float32))
    assert np.isinf(hi[0])
    rec = hi[1:].astype(np.float64) + lo[1:].astype(np.float64) / 2048.0
    assert np.abs(rec - np.array([1.0e-6, 3.0e-9])).max() < 2.0 ** -35
