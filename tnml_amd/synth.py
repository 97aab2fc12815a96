"""Synthetic inputs for the fixedL hot path: MNIST-shaped images and a random weight MPS.

The reference trains on MNIST idx-ubyte files (fixedL.cc:613 -> mllib/mnist.h:443-530), which are
absent offline; bench.py and the tests use this seeded generator instead
(SURVEY.md 8d).  The reference builds its initial W from sums of training product states with a
time-seeded RNG (fixedL.cc:702-728, util.h:104-121), so any deterministic initial W is an equally
valid starting point; `random_mps` mimics its shape: the s=0 ("pixel is black") component of every
site tensor is an isometry, so environments neither grow nor vanish along the chain.
"""
import numpy as np

NL = 10
# label histogram of MNIST train (mllib/MNIST/train-labels-idx1-ubyte)
MNIST_TRAIN_HIST = (5923, 6742, 5958, 6131, 5842, 5421, 5918, 6265, 5851, 5949)


def synthetic_labels(NT, seed=20160519, per_label=None):
    """Label sequence: `per_label` images of each label (reference semantics of Ntrain,
    mllib/mnist.h:472-496) or, if None, NT labels following the MNIST histogram, shuffled."""
    rng = np.random.default_rng(seed)
    if per_label is not None:
        lab = np.repeat(np.arange(NL, dtype=np.int32), per_label)
    else:
        hist = np.array(MNIST_TRAIN_HIST, dtype=np.float64)
        cnt = np.floor(hist / hist.sum() * NT).astype(np.int64)
        cnt[: NT - cnt.sum()] += 1
        lab = np.repeat(np.arange(NL, dtype=np.int32), cnt)
    rng.shuffle(lab)
    return lab.astype(np.int32)


def synthetic_images(N, labels, seed=20160519):
    """uint8 pixels [NT,N]: per label a fixed template of 3 Gaussian blobs on the sqrt(N) grid
    (peak 255) plus N(0,32^2) noise, clipped; outside the template support pixels are zeroed with
    probability 0.85 so that most pixels are 0 (MNIST-like sparsity)."""
    labels = np.asarray(labels)
    NT = labels.shape[0]
    side = int(round(np.sqrt(N)))
    rng = np.random.default_rng(seed + 1)
    yy, xx = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    templ = np.zeros((NL, side * side))
    for l in range(NL):
        t = np.zeros((side, side))
        for _ in range(3):
            cy, cx = rng.uniform(0.2 * side, 0.8 * side, size=2)
            sg = rng.uniform(0.06 * side, 0.14 * side)
            t += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
        templ[l] = 255.0 * t.ravel() / t.max()
    full = np.zeros((NL, N))
    full[:, : min(N, side * side)] = templ[:, : min(N, side * side)]
    out = np.empty((NT, N), dtype=np.uint8)
    chunk = 8192
    for s in range(0, NT, chunk):
        lab = labels[s : s + chunk]
        base = full[lab]
        noisy = base + rng.normal(0.0, 32.0, size=base.shape)
        kill = (base < 16.0) & (rng.random(base.shape) < 0.85)
        noisy[kill] = 0.0
        out[s : s + chunk] = np.clip(np.rint(noisy), 0, 255).astype(np.uint8)
    return out


def features_series(pixels, dtype=np.float64):
    """Reference feature map (fixedL.cc:637-642 after mllib/mnist.h:495): phi = [1, byte/260100]."""
    x = pixels.astype(np.float64) / 255.0 / 255.0
    return np.stack([np.ones_like(x), x / 4.0], axis=-1).astype(dtype)


def bond_dims(N, m):
    """dims[j] = dimension of the link between sites j and j+1 (j=0..N; dims[0]=dims[N]=1)."""
    return [1] + [int(min(m, 2 ** min(j, N - j, 20))) for j in range(1, N)] + [1]


def _isometry(rng, r, c):
    q, _ = np.linalg.qr(rng.standard_normal((max(r, c), min(r, c))))
    return q if r >= c else q.T


def random_mps(N, m, seed=1, s1_scale=1.0):
    """Deterministic random weight MPS, Label index (dim 10) on site N/2 (fixedL.cc:616,734).
    Returns a list of N arrays A_j[a,s,r] (A_{N/2}[a,s,r,l]) in ITensor index order."""
    rng = np.random.default_rng(seed)
    dims = bond_dims(N, m)
    c0 = N // 2
    W = []
    for j in range(1, N + 1):
        ml, mr = dims[j - 1], dims[j]
        if j == c0:
            A = np.empty((ml, 2, mr, NL))
            for l in range(NL):
                A[:, 0, :, l] = _isometry(rng, ml, mr) / np.sqrt(NL)
                A[:, 1, :, l] = s1_scale * rng.standard_normal((ml, mr)) / np.sqrt(NL * max(ml, mr))
        else:
            A = np.empty((ml, 2, mr))
            A[:, 0, :] = _isometry(rng, ml, mr)
            A[:, 1, :] = s1_scale * rng.standard_normal((ml, mr)) / np.sqrt(max(ml, mr))
        W.append(A)
    return W


def write_idx(dirname, pixels, labels, train=True, side=None):
    """Write images/labels as idx-ubyte files with the MNIST names the reference reads
    (mllib/mnist.h:244,262,279,297): big-endian header, magic 0x803 / 0x801."""
    import os
    import struct
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    n, npix = pixels.shape
    side = side or int(round(np.sqrt(npix)))
    rows, cols = (side, npix // side)
    assert rows * cols == npix
    stem = "train" if train else "t10k"
    os.makedirs(dirname, exist_ok=True)
    with open(os.path.join(dirname, stem + "-images-idx3-ubyte"), "wb") as f:
        f.write(struct.pack(">IIII", 0x803, n, rows, cols))
        f.write(pixels.tobytes())
    with open(os.path.join(dirname, stem + "-labels-idx1-ubyte"), "wb") as f:
        f.write(struct.pack(">II", 0x801, n))
        f.write(np.asarray(labels, dtype=np.uint8).tobytes())
