"""Numerics study (not a test): would Winograd F(2,3) along y AND x keep the 2e-5 per-kernel bar of the H3 conv?
Emulates, on the CPU in float32, the k3 conv of one 16^3 volume as (a) the direct form, (b) F(2,3) along x (what dm3d_conv_h3w.hip runs),
(c) F(2,3) x F(2,3) along y and x with the three dz taps direct — input / weight / output transforms in float32, operands rounded to the 22
significant bits the float16 hi + lo split keeps, float32 accumulation over (dz, Cin) — against float64.  usage: python tests/study_winograd2d.py"""
import numpy as np

rng = np.random.default_rng(0)


def r22(x):
    """float32 -> the value hi + lo of the float16 split represents (|x - hi - lo| <= 2^-22 |x|): keep 22 significant bits"""
    m, e = np.frexp(x.astype(np.float64))
    return (np.round(m * (1 << 22)) / (1 << 22) * np.exp2(e)).astype(np.float32)


def ref(x, w):          # x [D+2, H+2, W+2, C] (zero padded), w [3, 3, 3, C, K] -> [D, H, W, K] float64
    D, H, W = x.shape[0] - 2, x.shape[1] - 2, x.shape[2] - 2
    y = np.zeros((D, H, W, w.shape[-1]))
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                y += x[dz:dz + D, dy:dy + H, dx:dx + W].astype(np.float64) @ w[dz, dy, dx].astype(np.float64)
    return y


def direct(x, w):
    D, H, W = x.shape[0] - 2, x.shape[1] - 2, x.shape[2] - 2
    xs, ws = r22(x), r22(w)
    y = np.zeros((D, H, W, w.shape[-1]), np.float32)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                y += xs[dz:dz + D, dy:dy + H, dx:dx + W] @ ws[dz, dy, dx]
    return y


BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)      # v = B^T d
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float32)          # u = G g
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)                                   # y = A^T m


def wino_x(x, w):
    D, H, W = x.shape[0] - 2, x.shape[1] - 2, x.shape[2] - 2
    y = np.zeros((D, H, W, w.shape[-1]), np.float32)
    u = r22(np.einsum("tg,zygck->zytck", G, w))                                             # [3, 3, 4, C, K]
    for i in range(0, W, 2):
        v = r22(np.einsum("td,zydc->zytc", BT, x[:, :, i:i + 4]))                           # [D+2, H+2, 4, C]
        m = np.zeros((D, H, 4, w.shape[-1]), np.float32)
        for dz in range(3):
            for dy in range(3):
                m += np.einsum("zytc,tck->zytk", v[dz:dz + D, dy:dy + H], u[dz, dy])
        y[:, :, i:i + 2] = np.einsum("ot,zytk->zyok", AT, m)
    return y


def wino_yx(x, w):
    D, H, W = x.shape[0] - 2, x.shape[1] - 2, x.shape[2] - 2
    y = np.zeros((D, H, W, w.shape[-1]), np.float32)
    u = r22(np.einsum("sg,tq,zgqck->zstck", G, G, w))                                       # [3, 4, 4, C, K]
    for j in range(0, H, 2):
        for i in range(0, W, 2):
            d = x[:, j:j + 4, i:i + 4]
            v = r22(np.einsum("sa,tb,zabc->zstc", BT, BT, d).astype(np.float32))            # [D+2, 4, 4, C]
            m = np.zeros((D, 4, 4, w.shape[-1]), np.float32)
            for dz in range(3):
                m += np.einsum("zstc,stck->zstk", v[dz:dz + D], u[dz])
            y[:, j:j + 2, i:i + 2] = np.einsum("os,pt,zstk->zopk", AT, AT, m)
    return y


for C, K, scale in ((64, 64, 1.0), (192, 64, 1.0), (64, 64, 30.0)):
    x = np.zeros((18, 18, 18, C), np.float32)
    x[1:-1, 1:-1, 1:-1] = (rng.standard_normal((16, 16, 16, C)) * scale).astype(np.float32)
    x = x * (1.0 / (1.0 + np.exp(-x)))                                                      # post-swish activations
    w = (rng.standard_normal((3, 3, 3, C, K)) / np.sqrt(27 * C)).astype(np.float32)
    yr = ref(x, w)
    rel = lambda a: float(np.abs(a - yr).max() / np.abs(yr).max())
    print(f"Cin {C:3d} Cout {K} |x| ~ {scale:4.1f}: direct {rel(direct(x, w)):.2e}   F(2,3) along x {rel(wino_x(x, w)):.2e}   "
          f"F(2,3) x F(2,3) along y, x {rel(wino_yx(x, w)):.2e}   (bar 2e-5)")
