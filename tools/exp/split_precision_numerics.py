"""Numerics study (CPU, numpy): can the 3x3x3 convolution's fp32 GEMM run on the 16x faster 16-bit MFMA pipes of
gfx950 WITHOUT leaving fp32 accuracy, by splitting each fp32 operand into a short sum of 16-bit values and
accumulating the cross products in fp32?

Variants (cost = 16-bit MFMAs per fp32 MFMA worth of work; fp32 32x32x2 MFMA = 16 units, 16-bit 32x32x16 = 1 unit):
  fp32      native fp32 MFMA, fp32 accumulation in K order                                       cost 16
  bf16x9    a = a1+a2+a3 (bf16 each), all 9 products                                             cost  9
  bf16x6    same split, products with weight >= 2^-16 only (a1b1 a1b2 a2b1 a2b2 a1b3 a3b1)       cost  6
  fp16x2s   a = a1 + 2^-11 a2' (fp16 each, residual rescaled so it keeps 11 bits), a1b1 in one   cost  3
            accumulator, a1b2'+a2'b1 in a second one, combined at the end; operands pre-scaled
            by a power of two per tensor so that a1 stays in fp16's normal range
Error is measured against the float64 result of the fp32 inputs, relative to the output's rms, for a reduction of
K = 27 taps x 64 channels with activation / weight statistics like the PVConv layers (Swish outputs, N(0, 1/sqrt(K))
weights).  Prints one line per variant; `python tools/exp/split_precision_numerics.py`."""
import numpy as np

rs = np.random.RandomState(0)
M, N, K, KC = 512, 64, 27 * 64, 16


def bf16(x):  # round-to-nearest-even truncation of fp32 to 8 significant bits
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def mfma_accumulate(prods):
    """fp32 accumulation over K in chunks of KC (one MFMA each): exact inside a chunk, one rounding per chunk"""
    acc = np.zeros(prods[0][0].shape[:1] + prods[0][1].shape[1:], np.float32)
    for k0 in range(0, K, KC):
        part = sum(a[:, k0:k0 + KC].astype(np.float64) @ b[k0:k0 + KC].astype(np.float64) for a, b in prods)
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


x = rs.randn(M, K).astype(np.float32)
a = (x / (1 + np.exp(-x))).astype(np.float32)  # Swish outputs
a[rs.rand(M, K) < 0.3] = 0  # empty voxels
w = (rs.randn(K, N) / np.sqrt(K)).astype(np.float32)
truth = a.astype(np.float64) @ w.astype(np.float64)
scale = np.sqrt((truth ** 2).mean())


def report(name, y, cost):
    e = (y.astype(np.float64) - truth) / scale
    print(f"{name:8s} cost {cost:2d}/16   rms err {np.sqrt((e ** 2).mean()):.2e}   max err {np.abs(e).max():.2e}")


# native fp32: K chunk of 2 per MFMA
acc = np.zeros((M, N), np.float32)
for k0 in range(0, K, 2):
    acc = (acc.astype(np.float64) + a[:, k0:k0 + 2].astype(np.float64) @ w[k0:k0 + 2].astype(np.float64)).astype(np.float32)
report("fp32", acc, 16)

a1 = bf16(a); a2 = bf16(a - a1); a3 = bf16(a - a1 - a2)
w1 = bf16(w); w2 = bf16(w - w1); w3 = bf16(w - w1 - w2)
report("bf16x9", mfma_accumulate([(p, q) for p in (a1, a2, a3) for q in (w1, w2, w3)]), 9)
report("bf16x6", mfma_accumulate([(a1, w1), (a1, w2), (a2, w1), (a2, w2), (a1, w3), (a3, w1)]), 6)
report("bf16x3", mfma_accumulate([(a1, w1), (a1, w2), (a2, w1)]), 3)


def split16(t):
    s = 2.0 ** np.floor(np.log2(32768.0 / np.abs(t).max()))  # per-tensor power of two: max lands in [2^14, 2^15)
    hi = f16(t * s)
    lo = f16((t * s - hi) * 2048.0)
    return hi, lo, s


ah, al, sa = split16(a)
wh, wl, sw = split16(w)
main = mfma_accumulate([(ah, wh)])
corr = mfma_accumulate([(ah, wl), (al, wh)])
report("fp16x2s", ((main.astype(np.float64) + corr.astype(np.float64) / 2048.0) / (sa * sw)).astype(np.float32), 3)
corr2 = mfma_accumulate([(al, wl)])
report("fp16x4s", ((main.astype(np.float64) + corr.astype(np.float64) / 2048.0 + corr2.astype(np.float64) / 2048.0 ** 2)
                   / (sa * sw)).astype(np.float32), 4)

# ---- the variant a kernel would implement: NO per-tensor scale (hi = f16(a) directly; fp16 normal range is
# 6.1e-5 .. 65504, below it hi goes subnormal and the rescaled residual picks the rest up), on harder data ----
print("\nunscaled fp16 hi + 2^11-rescaled lo, against native fp32, on other operand statistics:")


def unscaled(a, w):
    ah = f16(a); al = f16((a - ah) * 2048.0)
    wh = f16(w); wl = f16((w - wh) * 2048.0)
    m = mfma_accumulate([(ah, wh)]); c = mfma_accumulate([(ah, wl), (al, wh)])
    return (m.astype(np.float64) + c.astype(np.float64) / 2048.0).astype(np.float32)


def native(a, w):
    acc = np.zeros((a.shape[0], w.shape[1]), np.float32)
    for k0 in range(0, K, 2):
        acc = (acc.astype(np.float64) + a[:, k0:k0 + 2].astype(np.float64) @ w[k0:k0 + 2].astype(np.float64)).astype(np.float32)
    return acc


cases = {
    "swish x N(0,1/sqrtK)": (a, w),
    "heavy tails (lognormal s=3)": ((rs.randn(M, K) * np.exp(3 * rs.randn(M, K))).astype(np.float32),
                                    (rs.randn(K, N) * np.exp(3 * rs.randn(K, N)) / np.sqrt(K)).astype(np.float32)),
    "tiny weights (1e-6)": (a, (w * 1e-6 / np.abs(w).mean()).astype(np.float32)),
    "large activations (1e3)": ((a * 1e3).astype(np.float32), w),
    "constant field + noise": ((3.0 + 1e-3 * rs.randn(M, K)).astype(np.float32), w),
}
for name, (aa, ww) in cases.items():
    t = aa.astype(np.float64) @ ww.astype(np.float64)
    s = np.sqrt((t ** 2).mean())
    e_em = np.sqrt((((unscaled(aa, ww) - t) / s) ** 2).mean())
    e_na = np.sqrt((((native(aa, ww) - t) / s) ** 2).mean())
    print(f"  {name:30s} fp16x2 rms {e_em:.2e}   native fp32 rms {e_na:.2e}   max|a| {np.abs(aa).max():.1e}")
