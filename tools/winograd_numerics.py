"""Round-5 verdict item 2, kill criterion 1 (CPU, numpy): Winograd F(2x2x2, 3x3x3) for the r <= 16 voxel convolutions
on the split pipe -- 27 -> 8 multiplies per output, transforms in fp32, the 64 per-position GEMMs [Cout x Cin] x
[Cin x tiles] as fp16 hi/lo products with fp32 accumulation (csrc/conv3d_split.hip's arithmetic) -- against a float64
direct convolution, on the layer the verdict names (128 -> 128 @ 16^3) and on the dynamic-range cases of
tests/test_conv_split_gpu.py::test_split_adversarial_dynamic_range.  Criterion: max error <= 5e-6 of the output's
maximum (the tests' BOUND).  The direct split kernel's own arithmetic is emulated beside it for scale.

  python tools/winograd_numerics.py        (~1 min)

Transforms (Lavin & Gray): B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],
A^T = [[1,1,1,0],[0,1,-1,-1]], applied along d, h, w."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
rs = np.random.RandomState(0)


def f16(a):
    with np.errstate(over="ignore"):
        return a.astype(np.float16).astype(np.float32)


def cut(a, scale_exp):
    """the kernel's cut after an exact power-of-two block scale: a 2^e = hi + lo / 2048"""
    s = np.float32(2.0) ** scale_exp
    v = (a * s).astype(np.float32)
    hi = f16(v)
    lo = f16((v - hi) * np.float32(2048))
    return hi, lo, s


def block_exp(a):
    m = np.abs(a[np.isfinite(a)]).max() if a.size else 0.0
    return 0 if m == 0 else 13 - int(np.floor(np.log2(m)))


def split_gemm(W, X):
    """[M, K] x [K, N] as the split kernel computes it: per-tensor weight scale, per-block activation scale, three fp16
    products accumulated in fp32 per K = 16 chunk (one rounding per chunk), D = main + corr / 2048"""
    wh, wl, ws = cut(W, block_exp(W))
    xh, xl, xs = cut(X, block_exp(X))
    main = np.zeros((W.shape[0], X.shape[1]), np.float32)
    corr = np.zeros_like(main)
    for k0 in range(0, W.shape[1], 16):
        a, b = slice(None), slice(k0, k0 + 16)
        main = (main.astype(np.float64) + wh[a, b].astype(np.float64) @ xh[b].astype(np.float64)).astype(np.float32)
        corr = (corr.astype(np.float64) + wh[a, b].astype(np.float64) @ xl[b].astype(np.float64)
                + wl[a, b].astype(np.float64) @ xh[b].astype(np.float64)).astype(np.float32)
    return ((main + corr * np.float32(1 / 2048)) / xs / ws).astype(np.float32)


def direct64(x, w):
    """x [Cin, r, r, r], w [Cout, Cin, 3, 3, 3] -> [Cout, r, r, r], pad 1, float64"""
    r = x.shape[1]
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (1, 1)))
    out = np.zeros((w.shape[0], r, r, r))
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                out += np.einsum("oc,cdhw->odhw", w[:, :, kd, kh, kw].astype(np.float64), xp[:, kd:kd + r, kh:kh + r, kw:kw + r])
    return out


def direct_split(x, w):
    """the direct kernel's arithmetic: K = Cin x 27 walked tap by tap, one block scale (emulated per whole tensor)"""
    r = x.shape[1]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1)))
    cols = np.stack([xp[:, kd:kd + r, kh:kh + r, kw:kw + r] for kd in range(3) for kh in range(3) for kw in range(3)], 1)
    X = cols.reshape(x.shape[0] * 27, -1).astype(np.float32)
    W = w.reshape(w.shape[0], -1).astype(np.float32)
    return split_gemm(W, X).reshape(w.shape[0], r, r, r)


def winograd(x, w, gemm):
    """F(2x2x2, 3x3x3): fp32 transforms (as VALU would), `gemm` per Winograd position"""
    cin, r = x.shape[0], x.shape[1]
    nt = r // 2
    xp = np.pad(x.astype(np.float32), ((0, 0), (1, 1), (1, 1), (1, 1)))
    U = np.einsum("ai,bj,ck,ocijk->abcoc", G, G, G, w.astype(np.float32), optimize=True).astype(np.float32) \
        if False else np.einsum("ai,bj,ck,onijk->abcon", G, G, G, w.astype(np.float32), optimize=True).astype(np.float32)
    # input tiles d[c, t, 4, 4, 4]
    idx = (np.arange(nt) * 2)[:, None] + np.arange(4)[None]          # [nt, 4]
    d = xp[:, idx][:, :, :, idx][:, :, :, :, :, idx]                 # [c, td, 4, th, 4, tw, 4]
    d = d.transpose(0, 1, 3, 5, 2, 4, 6).reshape(cin, nt ** 3, 4, 4, 4)
    V = np.einsum("ai,bj,ck,ctijk->abcct", BT, BT, BT, d, optimize=True).astype(np.float32) \
        if False else np.einsum("ai,bj,ck,ntijk->abcnt", BT, BT, BT, d, optimize=True).astype(np.float32)
    M = np.empty((4, 4, 4, w.shape[0], nt ** 3), np.float32)
    for a in range(4):
        for b in range(4):
            for c in range(4):
                M[a, b, c] = gemm(U[a, b, c], V[a, b, c])
    Y = np.einsum("ia,jb,kc,abcot->otijk", AT, AT, AT, M, optimize=True).astype(np.float32)   # [o, t, 2, 2, 2]
    Y = Y.reshape(w.shape[0], nt, nt, nt, 2, 2, 2).transpose(0, 1, 4, 2, 5, 3, 6).reshape(w.shape[0], r, r, r)
    return Y


def gemm32(W, X):
    return (W.astype(np.float32) @ X.astype(np.float32)).astype(np.float32)


def log_uniform(shape, lo, hi):
    return (np.exp(rs.uniform(np.log(lo), np.log(hi), shape)) * np.where(rs.rand(*shape) < 0.5, -1, 1)).astype(np.float32)


def cases(cin, cout, r):
    w = (rs.randn(cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)
    x = rs.randn(cin, r, r, r).astype(np.float32)
    yield "gaussian", x, w
    yield "swish-activations", (x / (1 + np.exp(-x))).astype(np.float32), w
    yield "nine-decades", log_uniform(x.shape, 1e-6, 1e4), w
    x2 = x.copy(); x2[cin // 2: cin // 2 + 16] *= 1e4; x2[:16] *= 1e-3
    yield "huge-one-chunk", x2, w
    k = rs.randint(-2048, 2048, x.shape).astype(np.float32)
    yield "residual-bits", ((1 + k * 2.0 ** -22) * np.where(rs.rand(*x.shape) < 0.5, -1, 1)).astype(np.float32), w
    sp = np.zeros_like(x); m = rs.rand(r, r, r) < 0.1; sp[:, m] = x[:, m]
    yield "voxelised (10 % occupied)", sp, w


if __name__ == "__main__":
    BOUND = 5e-6
    print("layer 128 -> 16 (of 128) @ 16^3, one sample; error = max |y - y64| / max |y64|   (bound %.0e)" % BOUND)
    print(f"{'case':28s} {'direct split':>13s} {'winograd fp32':>14s} {'winograd split':>15s}   verdict")
    worst = 0.0
    for name, x, w in cases(128, 16, 16):
        ref = direct64(x, w)
        sc = np.abs(ref).max()
        e_d = np.abs(direct_split(x, w) - ref).max() / sc
        e_w32 = np.abs(winograd(x, w, gemm32) - ref).max() / sc
        e_ws = np.abs(winograd(x, w, split_gemm) - ref).max() / sc
        worst = max(worst, e_ws)
        print(f"{name:28s} {e_d:13.2e} {e_w32:14.2e} {e_ws:15.2e}   {'ok' if e_ws <= BOUND else 'FAILS the bound'}")
    print("worst Winograd-on-the-split-pipe error: %.2e of max -> criterion 1 %s" % (worst, "met" if worst <= BOUND else "NOT met"))
