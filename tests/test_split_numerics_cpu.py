"""The arithmetic of the split-operand kernels (csrc/split_ops.h) restated in numpy and held to its claim on the CPU:
an fp32 operand a is cut into two fp16 pieces a = a_h + a_l / 2048 after an exact power-of-two block scaling, the GEMM
is main = A_h B_h and corr = A_h B_l + A_l B_h accumulated in fp32, D = main + corr / 2048 -- and the result is as close
to the float64 product as an fp32 fmaf chain is, over ranges where a plain fp16 cast overflows, flushes or loses the
residual.  (The GPU tests hold the kernels themselves to the same bounds: test_conv_split_gpu.py,
test_pwconv_split_gpu.py.)"""
import numpy as np
import pytest


def scale_exp(m):
    """e with 2^13 <= m * 2^e < 2^14 (split_ops.h::scale_exp)"""
    return 13 - int(np.floor(np.log2(m)))


def cut(v):
    """split_ops.h::cut: hi = fp16(v), lo = fp16((v - hi) * 2048), both round-to-nearest-even like v_cvt_f16_f32"""
    v = v.astype(np.float32)
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def split_gemm(W, X, headroom=0):
    """W [M,K] (one scale per tensor), X [K,N] (one scale per column and 16-row chunk, monotone along K) -> [M,N]"""
    M, K = W.shape
    N = X.shape[1]
    ew = scale_exp(np.abs(W).max())
    Wh, Wl = cut(W * np.float32(2.0 ** ew))
    main = np.zeros((M, N), np.float32)
    corr = np.zeros((M, N), np.float32)
    E = np.full(N, 127)
    for k0 in range(0, K, 16):
        xc = X[k0:k0 + 16]
        m = np.abs(xc).max(0)
        for n in np.nonzero(m > 0)[0]:
            e = scale_exp(m[n])
            if e < E[n]:
                if E[n] != 127:
                    f = np.float32(2.0 ** (e - headroom - E[n]))
                    main[:, n] *= f
                    corr[:, n] *= f
                E[n] = e - headroom
        xs = np.where(E == 127, 1.0, 2.0 ** E.astype(np.float64)).astype(np.float32)
        Xh, Xl = cut(xc * xs[None, :])
        wh, wl = Wh[:, k0:k0 + 16].astype(np.float32), Wl[:, k0:k0 + 16].astype(np.float32)
        xh, xl = Xh.astype(np.float32), Xl.astype(np.float32)
        main += wh @ xh                      # fp32 accumulation (the MFMA's; summation order is not part of the claim)
        corr += wh @ xl + wl @ xh
    us = np.where(E == 127, 1.0, 2.0 ** (-E.astype(np.float64))).astype(np.float32)
    return ((main + corr * np.float32(1.0 / 2048.0)) * us[None, :]) * np.float32(2.0 ** -ew)


def rel_err(got, ref, per_column):
    d = np.abs(got.astype(np.float64) - ref)
    if per_column:
        return (d.max(0) / np.maximum(np.abs(ref).max(0), 1e-300)).max()
    return d.max() / np.abs(ref).max()


CASES = ["normal", "nine-decades", "beyond-fp16-max", "tiny", "per-column-scales", "huge-one-chunk", "tiny-weights",
         "huge-weights", "residual-bits"]


@pytest.mark.parametrize("headroom", [0, 4])
@pytest.mark.parametrize("case", CASES)
def test_split_product_is_fp32_accurate(case, headroom):
    rng = np.random.default_rng(7)
    M, K, N = 48, 96, 64
    W = rng.standard_normal((M, K)).astype(np.float32)
    X = rng.standard_normal((K, N)).astype(np.float32)
    per_column = False
    if case == "nine-decades":
        X = (np.exp(rng.uniform(np.log(1e-6), np.log(1e4), X.shape)) * rng.choice([-1.0, 1.0], X.shape)).astype(np.float32)
    elif case == "beyond-fp16-max":
        X *= np.float32(3.0e6)
    elif case == "tiny":
        X *= np.float32(1e-30)
    elif case == "per-column-scales":
        X *= np.exp(rng.uniform(np.log(1e-6), np.log(1e7), (1, N))).astype(np.float32)
        per_column = True
    elif case == "huge-one-chunk":
        X[32:48] *= np.float32(1e4)
        X[:16] *= np.float32(1e-3)
    elif case == "tiny-weights":
        W *= np.float32(1e-9)
    elif case == "huge-weights":
        W *= np.float32(1e6)
    elif case == "residual-bits":            # values whose low piece alone carries the information: 1 + k * 2^-22
        k = rng.integers(-2048, 2048, X.shape).astype(np.float32)
        X = ((1.0 + k * 2.0 ** -22) * rng.choice([-1.0, 1.0], X.shape)).astype(np.float32)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    got = split_gemm(W, X, headroom)
    assert np.isfinite(got).all()
    e = rel_err(got, ref, per_column)
    f32 = rel_err((W @ X).astype(np.float32), ref, per_column)       # numpy's fp32 GEMM as the yardstick
    assert e < 1e-6, (case, e, f32)
    assert e < 4.0 * f32 + 3e-7, (case, e, f32)                        # the same class as an fp32 chain


def test_plain_fp16_would_fail_where_the_split_does_not():
    """the control: a single fp16 cast of the same operands loses 3 decimal digits, overflows beyond 65504 and flushes
    small values -- the split with block scaling does none of it"""
    rng = np.random.default_rng(1)
    W = rng.standard_normal((32, 64)).astype(np.float32)
    X = rng.standard_normal((64, 32)).astype(np.float32)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    naive = W.astype(np.float16).astype(np.float32) @ X.astype(np.float16).astype(np.float32)
    assert rel_err(naive, ref, False) > 1e-4 and rel_err(split_gemm(W, X), ref, False) < 1e-6
    with np.errstate(over="ignore"):
        assert not np.isfinite((X * np.float32(3e6)).astype(np.float16)).all()
    assert ((X * np.float32(1e-30)).astype(np.float16) == 0).all()


def test_power_of_two_scaling_is_exact():
    """split(W, X * 2^k) == split(W, X) * 2^k bit for bit: no range in which the cut changes its rounding"""
    rng = np.random.default_rng(3)
    W = rng.standard_normal((32, 48)).astype(np.float32)
    X = rng.standard_normal((48, 16)).astype(np.float32)
    base = split_gemm(W, X, 4)
    for k in (-80, -17, 9, 40, 80):
        f = np.float32(2.0 ** k)
        assert np.array_equal(split_gemm(W, X * f, 4), base * f), k
        assert np.array_equal(split_gemm(W * f, X, 4), base * f), k
