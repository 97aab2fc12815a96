import sys, glob, numpy as np
for fn in sorted(glob.glob(sys.argv[1] + "*.bin")):
    hb = np.fromfile(fn)
    n, mk = int(hb[0]), int(hb[1])
    D = hb[2:2+n]; E = hb[2+n:2+2*n]; W = hb[2+2*n:2+3*n]; Z = hb[2+3*n:2+3*n+n*mk].reshape(mk, n).T
    T = np.diag(D) + np.diag(E[:n-1], 1) + np.diag(E[:n-1], -1)
    ev = np.linalg.eigvalsh(T)
    print(fn, "n", n, "mk", mk, "|T|", np.abs(ev).max())
    print("  eigenvalue err vs lapack (rel to max):", np.abs(np.sort(W) - ev).max() / np.abs(ev).max())
    top = ev[::-1][:mk]
    print("  top-mk eigenvalues: max %.3e  min %.3e ; smallest rel gaps among kept:" % (top[0], top[-1]), np.sort(np.abs(np.diff(top)) / top[0])[:6])
    S = Z.T @ Z - np.eye(mk)
    dev = np.abs(S).max(axis=0)
    bad = np.flatnonzero(dev > 1e-6)
    print("  max dev %.3e; #columns with dev>1e-6: %d; indices (largest-first order):" % (np.abs(S).max(), len(bad)), bad[:40])
    print("  eigenvalues (rel) of bad columns:", (top[bad] / top[0])[:20])
    res = np.abs(T @ Z - Z * top[None, :]).max(axis=0) / top[0]
    print("  residual max %.2e (bad cols: %.2e)" % (res.max(), res[bad].max() if len(bad) else 0))
    # coupling magnitudes
    print("  small |E|/|T| count (<1e-12):", (np.abs(E[:n-1]) < 1e-12 * np.abs(ev).max()).sum())
