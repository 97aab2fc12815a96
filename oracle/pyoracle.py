"""ctypes front-end of the C oracle (oracle/libfixedl_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Thin, one method per C entry point of oracle/fixedl_oracle.h.  numpy arrays cross the boundary
in ITensor order (first index fastest): A_j[a,s,r(,L)], B[a,s,t,r(,L)] as Fortran-ordered
arrays; environments [m(,L)].
"""
import ctypes as C
import os
import subprocess
import numpy as np

NL = 10
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libfixedl_oracle.so")
    src = os.path.join(_HERE, "fixedl_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class CgTrace(C.Structure):
    _fields_ = [("npass_done", C.c_int), ("converged", C.c_int), ("cost", C.c_double * 64),
                ("rnorm", C.c_double * 64), ("pAp", C.c_double * 64), ("alpha", C.c_double * 64)]


class BondReport(C.Structure):
    _fields_ = [("sweep", C.c_int), ("half", C.c_int), ("bond", C.c_int), ("c", C.c_int),
                ("origm", C.c_int), ("newm", C.c_int), ("truncerr", C.c_double),
                ("norm_newB", C.c_double), ("diff_B_newB", C.c_double),
                ("cost_after_svd", C.c_double), ("label_cost", C.c_double * NL),
                ("reg_cost", C.c_double), ("ncorrect", C.c_long), ("cg", CgTrace)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, dp, ip, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_features_series.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_ubyte), dp]
        L.orc_set_site.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.orc_site_dims.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.orc_get_site.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_init.argtypes = [C.c_void_p]
        L.orc_set_bond.argtypes = [C.c_void_p, C.c_int]
        L.orc_shiftE.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_env_dims.argtypes = [C.c_void_p, C.c_int, ip, ip]
        L.orc_get_env.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.orc_bond_dims.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.orc_bond_tensor.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_forward.argtypes = [C.c_void_p, dp, dp]
        L.orc_gradient.argtypes = [C.c_void_p, dp, dp]
        L.orc_quadcost.restype = C.c_double
        L.orc_quadcost.argtypes = [C.c_void_p, dp, C.c_double, dp, dp, C.POINTER(C.c_long)]
        L.orc_cgrad.argtypes = [C.c_void_p, dp, C.c_int, C.c_double, C.c_double, C.POINTER(CgTrace)]
        L.orc_truncate.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_double, dp]
        L.orc_svd_split.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                    dp, ip, dp, ip]
        L.orc_sweepnext.argtypes = [ip, ip, C.c_int]
        L.orc_mldmrg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                 C.c_double, C.c_int, C.POINTER(BondReport), C.c_int]
        L.orc_toverlap.argtypes = [C.c_void_p, C.c_int, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    """flatten in ITensor (first-index-fastest) order"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


class Oracle:
    """One TrainStates + W pair of the reference (fixedL.cc:64-274, :669-728)."""

    def __init__(self, phi, labels, W=None, nthread=1, nbatch=1):
        phi = np.ascontiguousarray(phi, dtype=np.float64)          # [NT,N,2]
        self.NT, self.N, d = phi.shape
        assert d == 2
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        self.labels = lab
        self.c0 = self.N // 2
        self._L = lib()
        self._h = self._L.orc_create(self.N, self.NT, _dp(phi), lab.ctypes.data_as(C.POINTER(C.c_int)),
                                     nthread, nbatch)
        if not self._h:
            raise ValueError(self._L.orc_last_error().decode())
        if W is not None:
            self.set_mps(W)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self._L.orc_last_error().decode())

    def set_site(self, j, A):
        A = np.asarray(A, dtype=np.float64)
        self._ck(self._L.orc_set_site(self._h, j, A.shape[0], A.shape[2], int(A.ndim == 4), _dp(_f(A))))

    def set_mps(self, W):
        for j, A in enumerate(W, start=1):
            self.set_site(j, A)

    def get_site(self, j):
        ml, mr, hl = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.orc_site_dims(self._h, j, ml, mr, hl))
        shape = (ml.value, 2, mr.value) + ((NL,) if hl.value else ())
        buf = np.empty(int(np.prod(shape)))
        self._ck(self._L.orc_get_site(self._h, j, _dp(buf)))
        return buf.reshape(shape, order="F")

    def get_mps(self):
        return [self.get_site(j) for j in range(1, self.N + 1)]

    def init(self):
        self._ck(self._L.orc_init(self._h))

    def set_bond(self, b):
        self._ck(self._L.orc_set_bond(self._h, b))

    def shiftE(self, b, from_left):
        self._ck(self._L.orc_shiftE(self._h, b, int(bool(from_left))))

    def env(self, j):
        m, hl = C.c_int(), C.c_int()
        self._ck(self._L.orc_env_dims(self._h, j, m, hl))
        shape = (m.value,) + ((NL,) if hl.value else ())
        out = np.empty((self.NT,) + shape)
        buf = np.empty(int(np.prod(shape)))
        for i in range(self.NT):
            self._ck(self._L.orc_get_env(self._h, j, i, _dp(buf)))
            out[i] = buf.reshape(shape, order="F")
        return out

    def bond_shape(self, b):
        mL, mR, lab = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.orc_bond_dims(self._h, b, mL, mR, lab))
        return (mL.value, 2, 2, mR.value) + ((NL,) if lab.value else ())

    def bond_tensor(self, b):
        shape = self.bond_shape(b)
        buf = np.empty(int(np.prod(shape)))
        self._ck(self._L.orc_bond_tensor(self._h, b, _dp(buf)))
        return buf.reshape(shape, order="F")

    def forward(self, B):
        P = np.empty((self.NT, NL))
        self._ck(self._L.orc_forward(self._h, _dp(_f(B)), _dp(P)))
        return P

    def gradient(self, B):
        G = np.empty(B.size)
        self._ck(self._L.orc_gradient(self._h, _dp(_f(B)), _dp(G)))
        return G.reshape(B.shape, order="F")

    def quadcost(self, B, lam):
        lc = np.empty(NL)
        cr, nc = C.c_double(), C.c_long()
        Cst = self._L.orc_quadcost(self._h, _dp(_f(B)), lam, _dp(lc), C.byref(cr), C.byref(nc))
        return Cst, lc, cr.value, nc.value

    def cgrad(self, B, npass, lam, cconv):
        buf = _f(B).copy()                                         # cgrad updates B in place
        tr = CgTrace()
        self._ck(self._L.orc_cgrad(self._h, _dp(buf), npass, lam, cconv, C.byref(tr)))
        n = tr.npass_done
        trace = dict(npass_done=n, converged=bool(tr.converged), cost=list(tr.cost[:max(n - 1, 0) if not tr.converged else n]),
                     rnorm=list(tr.rnorm[:max(n - 1, 0) if not tr.converged else n]),
                     pAp=list(tr.pAp[:n]), alpha=list(tr.alpha[:n]))
        return buf.reshape(B.shape, order="F"), trace

    def svd_split(self, B, b, ha, cutoff, maxm, minm):
        te, m, nsv = C.c_double(), C.c_int(), C.c_int()
        sv = np.empty(B.size)
        self._ck(self._L.orc_svd_split(self._h, _dp(_f(B)), b, ha, cutoff, maxm, minm, C.byref(te), C.byref(m),
                                       _dp(sv), C.byref(nsv)))
        return m.value, te.value, sv[:nsv.value].copy()

    def mldmrg(self, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds=0, verbose=False):
        cap = max_bonds if max_bonds > 0 else nsweep * 2 * (self.N - 1)
        reps = (BondReport * cap)()
        n = self._L.orc_mldmrg(self._h, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds, reps, int(verbose))
        if n < 0:
            raise RuntimeError(self._L.orc_last_error().decode())
        out = []
        for r in reps[:n]:
            k = r.cg.npass_done
            out.append(dict(sweep=r.sweep, half=r.half, bond=r.bond, c=r.c, origm=r.origm, newm=r.newm,
                            truncerr=r.truncerr, norm_newB=r.norm_newB, diff=r.diff_B_newB, cost=r.cost_after_svd,
                            label_cost=np.array(r.label_cost[:]), reg_cost=r.reg_cost, ncorrect=r.ncorrect,
                            cg=dict(npass_done=k, converged=bool(r.cg.converged), cost=list(r.cg.cost[:k]),
                                    rnorm=list(r.cg.rnorm[:k]), pAp=list(r.cg.pAp[:k]), alpha=list(r.cg.alpha[:k]))))
        return out

    def toverlap(self, i):
        out = np.empty(NL)
        self._ck(self._L.orc_toverlap(self._h, i, _dp(out)))
        return out


def truncate(p, maxm, minm, cutoff):
    p = np.ascontiguousarray(p, dtype=np.float64)
    te = C.c_double()
    m = lib().orc_truncate(_dp(p), len(p), maxm, minm, cutoff, C.byref(te))
    return m, te.value


def sweepnext(b, ha, N):
    bb, hh = C.c_int(b), C.c_int(ha)
    lib().orc_sweepnext(C.byref(bb), C.byref(hh), N)
    return bb.value, hh.value


def features_series(pixels):
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    NT, N = pixels.shape
    phi = np.empty((NT, N, 2))
    lib().orc_features_series(N, NT, pixels.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(phi))
    return phi


# ---------------------------------------------------------------------------------------------------
# per-label variant (reference single.cc / single.h) -- oracle/single_oracle.c
class SingleBondReport(C.Structure):
    _fields_ = [("sweep", C.c_int), ("half", C.c_int), ("c", C.c_int), ("origm", C.c_int), ("newm", C.c_int),
                ("truncerr", C.c_double), ("cost_old", C.c_double), ("cost_cg", C.c_double), ("reg_cost", C.c_double),
                ("cost_after_svd", C.c_double), ("norm_oB", C.c_double), ("norm_newB", C.c_double),
                ("cg_skipped", C.c_int), ("cg", CgTrace)]


_S_BOUND = False


def _slib():
    global _S_BOUND
    L = lib()
    if not _S_BOUND:
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.sorc_create.restype = vp
        L.sorc_create.argtypes = [C.c_int, C.c_int, dp, ip, C.c_int, C.c_int]
        L.sorc_destroy.argtypes = [vp]
        L.sorc_features.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, dp]
        L.sorc_set_site.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
        L.sorc_site_dims.argtypes = [vp, C.c_int, ip, ip]
        L.sorc_get_site.argtypes = [vp, C.c_int, dp]
        L.sorc_init.argtypes = [vp]
        L.sorc_set_bond.argtypes = [vp, C.c_int]
        L.sorc_shiftE.argtypes = [vp, C.c_int, C.c_int]
        L.sorc_get_env.argtypes = [vp, C.c_int, C.c_int, dp, ip]
        L.sorc_bond_dims.argtypes = [vp, C.c_int, ip, ip]
        L.sorc_bond_tensor.argtypes = [vp, C.c_int, dp]
        L.sorc_forward.argtypes = [vp, dp, dp]
        L.sorc_gradient.argtypes = [vp, dp, dp]
        L.sorc_quadcost.restype = C.c_double
        L.sorc_quadcost.argtypes = [vp, dp, C.c_double, dp]
        L.sorc_cgrad.argtypes = [vp, dp, C.c_int, C.c_double, C.c_double, C.POINTER(CgTrace)]
        L.sorc_fast_cgrad.argtypes = [vp, dp, C.c_int, C.c_double, C.c_double, C.POINTER(CgTrace)]
        L.sorc_set_method.argtypes = [vp, C.c_int]
        L.sorc_set_pcut.argtypes = [vp, C.c_double]
        L.sorc_set_noise.argtypes = [vp, C.c_double]
        L.sorc_pinv.argtypes = [vp, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, ip, dp]
        L.sorc_noise_split.argtypes = [vp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp, ip]
        L.sorc_exact.argtypes = [vp, dp, C.c_double, C.c_double]
        L.sorc_svd_split.argtypes = [vp, dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, dp, ip, dp, ip]
        L.sorc_mldmrg.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.POINTER(SingleBondReport)]
        L.sorc_output.argtypes = [vp, C.c_int, dp]
        _S_BOUND = True
    return L


def features_single(pixels, normal=True):
    """single.cc:71-84: [cos(pi x/2), sin(pi x/2)] (normal) or [1, x/4] (series), x = (byte/255)/255"""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    NT, N = pixels.shape
    phi = np.empty((NT, N, 2))
    _slib().sorc_features(N, NT, pixels.ctypes.data_as(C.POINTER(C.c_ubyte)), int(normal), _dp(phi))
    return phi


class SingleOracle:
    """training states + plain weight MPS of the per-label variant (single.cc:153-218, single.h)"""

    def __init__(self, phi, labels, target, W=None, nthread=1):
        phi = np.ascontiguousarray(phi, dtype=np.float64)
        self.NT, self.N, d = phi.shape
        assert d == 2
        self.labels = np.ascontiguousarray(labels, dtype=np.int32)
        self.target = int(target)
        self._L = _slib()
        self._h = self._L.sorc_create(self.N, self.NT, _dp(phi), self.labels.ctypes.data_as(C.POINTER(C.c_int)), self.target, nthread)
        if not self._h:
            raise ValueError("sorc_create failed")
        if W is not None:
            self.set_mps(W)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.sorc_destroy(self._h)
            self._h = None

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError("single oracle call failed")
        return rc

    def set_site(self, j, A):
        A = np.asarray(A, dtype=np.float64)
        assert A.ndim == 3
        self._ck(self._L.sorc_set_site(self._h, j, A.shape[0], A.shape[2], _dp(_f(A))))

    def set_mps(self, W):
        for j, A in enumerate(W, start=1):
            self.set_site(j, A)

    def get_site(self, j):
        ml, mr = C.c_int(), C.c_int()
        self._ck(self._L.sorc_site_dims(self._h, j, ml, mr))
        buf = np.empty(ml.value * 2 * mr.value)
        self._ck(self._L.sorc_get_site(self._h, j, _dp(buf)))
        return buf.reshape((ml.value, 2, mr.value), order="F")

    def get_mps(self):
        return [self.get_site(j) for j in range(1, self.N + 1)]

    def init(self):
        self._ck(self._L.sorc_init(self._h))

    def set_bond(self, b):
        self._ck(self._L.sorc_set_bond(self._h, b))

    def shiftE(self, b, from_left):
        self._ck(self._L.sorc_shiftE(self._h, b, int(bool(from_left))))

    def env(self, j):
        m = C.c_int()
        self._ck(self._L.sorc_get_env(self._h, j, 0, None, m))
        out = np.empty((self.NT, m.value))
        for i in range(self.NT):
            self._ck(self._L.sorc_get_env(self._h, j, i, _dp(out[i]), None))
        return out

    def bond_shape(self, b):
        mL, mR = C.c_int(), C.c_int()
        self._ck(self._L.sorc_bond_dims(self._h, b, mL, mR))
        return (mL.value, 2, 2, mR.value)

    def bond_tensor(self, b):
        shape = self.bond_shape(b)
        buf = np.empty(int(np.prod(shape)))
        self._ck(self._L.sorc_bond_tensor(self._h, b, _dp(buf)))
        return buf.reshape(shape, order="F")

    def forward(self, B):
        P = np.empty(self.NT)
        self._ck(self._L.sorc_forward(self._h, _dp(_f(B)), _dp(P)))
        return P

    def gradient(self, B):
        B = np.asarray(B)
        G = np.empty(B.size)
        self._ck(self._L.sorc_gradient(self._h, _dp(_f(B)), _dp(G)))
        return G.reshape(B.shape, order="F")

    def quadcost(self, B, lam):
        cr = C.c_double()
        c = self._L.sorc_quadcost(self._h, _dp(_f(B)), lam, C.byref(cr))
        return c, cr.value

    def cgrad(self, B, npass, lam, cconv):
        B = np.asarray(B)
        buf = _f(B).copy()
        tr = CgTrace()
        skipped = self._ck(self._L.sorc_cgrad(self._h, _dp(buf), npass, lam, cconv, C.byref(tr)))
        n = tr.npass_done
        return buf.reshape(B.shape, order="F"), dict(skipped=bool(skipped), npass_done=n, converged=bool(tr.converged), cost=list(tr.cost[:max(n - 1, 0)] if not tr.converged else tr.cost[:n]),
                                                     rnorm=list(tr.rnorm[:max(n - 1, 0)] if not tr.converged else tr.rnorm[:n]), pAp=list(tr.pAp[:n]), alpha=list(tr.alpha[:n]))

    def fast_cgrad(self, B, npass, lam, cconv):
        """single.h:290-398 (method = fast_conj): no cost in the trace"""
        B = np.asarray(B)
        buf = _f(B).copy()
        tr = CgTrace()
        skipped = self._ck(self._L.sorc_fast_cgrad(self._h, _dp(buf), npass, lam, cconv, C.byref(tr)))
        n = tr.npass_done
        return buf.reshape(B.shape, order="F"), dict(skipped=bool(skipped), npass_done=n, converged=bool(tr.converged), cost=[],
                                                     rnorm=list(tr.rnorm[:max(n - 1, 0)] if not tr.converged else tr.rnorm[:n]), pAp=list(tr.pAp[:n]), alpha=list(tr.alpha[:n]))

    def exact(self, b, lam, pcut=1e-8):
        """single.h:117-160 (method = exact) for the bond set by set_bond(b): returns the solved bond tensor"""
        buf = np.zeros(int(np.prod(self.bond_shape(b))))
        self._ck(self._L.sorc_exact(self._h, _dp(buf), lam, pcut))
        return buf.reshape(self.bond_shape(b), order="F")

    def set_method(self, method, pcut=1e-8):
        """optimiser of mldmrg: "conj" (cgrad), "fast_conj" (fast_cgrad) or "exact", single.h:598-600"""
        self._ck(self._L.sorc_set_method(self._h, {"conj": 0, "fast_conj": 1, "exact": 2}[method]))
        self._ck(self._L.sorc_set_pcut(self._h, pcut))

    def pinv(self, b, V0, npass, lam, pcut=1e-8):
        """single.h:404-517 from the start V0 [D, r] (columns = r tensors in ITensor order, flattened column-major); returns (B, trace of V*E, D)"""
        V0 = np.asfortranarray(V0, dtype=np.float64)
        D, r = V0.shape
        assert D == int(np.prod(self.bond_shape(b)))
        B = np.zeros(D); ve = np.zeros(npass + 1); Dsv = np.zeros(r); done = C.c_int()
        self._ck(self._L.sorc_pinv(self._h, _dp(V0), r, npass, lam, pcut, _dp(B), _dp(ve), C.byref(done), _dp(Dsv)))
        return B.reshape(self.bond_shape(b), order="F"), ve[:done.value + 1].copy(), Dsv

    def set_noise(self, noise):
        """sweeps.noise() of single.cc:25,222: >= 1e-14 makes mldmrg split through rho + noise * drho (single.h:648-672)"""
        self._ck(self._L.sorc_set_noise(self._h, noise))

    def noise_split(self, B, b, ha, noise, cutoff, maxm, minm):
        te, m = C.c_double(), C.c_int()
        self._ck(self._L.sorc_noise_split(self._h, _dp(_f(B)), b, ha, noise, cutoff, maxm, minm, C.byref(te), C.byref(m)))
        return m.value, te.value

    def svd_split(self, B, b, ha, cutoff, maxm, minm):
        te, m, nsv = C.c_double(), C.c_int(), C.c_int()
        sv = np.zeros(4 * max(self.bond_shape(b)[0], self.bond_shape(b)[3]))
        self._ck(self._L.sorc_svd_split(self._h, _dp(_f(B)), b, ha, cutoff, maxm, minm, C.byref(te), C.byref(m), _dp(sv), C.byref(nsv)))
        return m.value, te.value, sv[:nsv.value].copy()

    def mldmrg(self, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds=0):
        cap = max_bonds if max_bonds > 0 else nsweep * 2 * (self.N - 1)
        reps = (SingleBondReport * cap)()
        n = self._ck(self._L.sorc_mldmrg(self._h, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds, reps))
        out = []
        for r in reps[:n]:
            k = r.cg.npass_done
            out.append(dict(sweep=r.sweep, half=r.half, c=r.c, origm=r.origm, newm=r.newm, truncerr=r.truncerr,
                            cost_old=r.cost_old, cost_cg=r.cost_cg, reg_cost=r.reg_cost, cost=r.cost_after_svd,
                            norm_oB=r.norm_oB, norm_newB=r.norm_newB, cg_skipped=bool(r.cg_skipped),
                            cg_cost=list(r.cg.cost[:max(k - 1, 0)]), cg_alpha=list(r.cg.alpha[:k]), cg_rnorm=list(r.cg.rnorm[:max(k - 1, 0)])))
        return out

    def output(self, i):
        f = C.c_double()
        self._ck(self._L.sorc_output(self._h, i, C.byref(f)))
        return f.value
