"""Go / no-go numerics for a Winograd F(2x2, 3x3) form of the split-fp16 convolution, on CPU in numpy.

Both forms use the product's arithmetic model: operands split into fp16 (hi, lo) pairs after a power-of-two scaling, three
products (hi*hi + lo*hi + hi*lo) accumulated in fp32.  The Winograd form transforms the fp32 activations (B^T d B, additions
only) and the weights (G g G^T, in fp64, rounded once) BEFORE the split, multiplies per transform position, and applies
A^T m A in fp32.  Reported: max error against an fp64 direct convolution, relative to max|y| (the conv tests' measure)."""
import numpy as np

rng = np.random.default_rng(0)


def split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def mm3(a, b):
    """sum_k a[m,k] b[k,n] with the three-pass split product, fp32 accumulation (a: weights, scaled into fp16's range)."""
    amax = np.abs(a).max()
    e = 12 - int(np.floor(np.log2(amax)))
    ah, al = split(a * np.float32(2.0 ** e))
    bh, bl = split(b)
    acc = ah @ bh
    acc = acc + al @ bh
    acc = acc + ah @ bl
    return acc * np.float32(2.0 ** -e)


def direct(x, w):
    C, H, W = x.shape
    K = w.shape[0]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    cols = np.stack([xp[:, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], 1).reshape(C * 9, H * W)
    return cols, w.reshape(K, C * 9)


Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd(x, w):
    C, H, W = x.shape
    K = w.shape[0]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(np.float32)
    th, tw = H // 2, W // 2
    # tiles d[c, ty, tx, 4, 4]
    d = np.stack([np.stack([xp[:, i:i + 2 * th:2, j:j + 2 * tw:2] for j in range(4)], -1) for i in range(4)], -2)
    Bt32 = Bt.astype(np.float32)
    V = np.einsum("ij,ctxjk,lk->ctxil", Bt32, d, Bt32).astype(np.float32)          # additions only: exact-ish in fp32
    U = np.einsum("ij,kcjl,ml->kcim", G, w.astype(np.float64), G).astype(np.float32)
    M = np.empty((K, th, tw, 4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            M[:, :, :, i, j] = mm3(U[:, :, i, j], V[:, :, :, i, j].reshape(C, th * tw)).reshape(K, th, tw)
    At32 = At.astype(np.float32)
    Y = np.einsum("ij,ktxjl,ml->ktxim", At32, M, At32).astype(np.float32)            # (K, th, tw, 2, 2)
    return Y.transpose(0, 1, 3, 2, 4).reshape(K, H, W)


for C, K, H, W, relu_in in ((128, 64, 32, 48, False), (384, 128, 32, 48, False), (384, 128, 32, 48, True), (64, 64, 48, 64, True)):
    x = rng.standard_normal((C, H, W)).astype(np.float32)
    if relu_in:
        x = np.maximum(x, 0)
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    cols, wm = direct(x, w)
    ref = wm.astype(np.float64) @ cols.astype(np.float64)
    yd = mm3(wm, cols)
    yw = winograd(x, w).reshape(K, H * W)
    s = np.abs(ref).max()
    print("Cin %3d Cout %3d %dx%d relu_in=%d: direct split-fp16 %.2e   winograd F(2x2,3x3) split-fp16 %.2e   (x%.1f)"
          % (C, K, H, W, relu_in, np.abs(yd - ref).max() / s, np.abs(yw - ref).max() / s,
             np.abs(yw - ref).max() / np.abs(yd - ref).max()))
