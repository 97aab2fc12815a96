"""Host-side mirror of the reference's fixedL interface on top of the C-ABI (include/tnml.h).

Names follow /root/reference/fixedL.cc so that parity tests read like the reference:
  TrainStates.{init,setBond,shiftE}   fixedL.cc:122-233   (device resident environments)
  quadcost / cgrad                    fixedL.cc:280-445
  mldmrg                              fixedL.cc:451-570   (sweep loop; one C call per bond update)
Tensors cross this layer as numpy arrays with ITensor index order as axis order:
A_j[a,s,r(,L)], B[a,s,t,r(,L)], E[n,m(,L)].  Everything here calls the HIP path; nothing falls
back to the CPU and nothing imports the oracle.
"""
import ctypes as C

import numpy as np

from . import lib as _lib

NL = _lib.NL


class TnmlError(RuntimeError):
    pass


class TrainStates:
    """Training set + environments + W replica of one rank (TrainStates + MPS W of fixedL.cc)."""

    def __init__(self, labels, N, maxm, pixels=None, phi=None, device=0, rank=0, nranks=1, NT_total=None, dtype="f64",
                 single_label=None, svd_backend=0):
        """single_label = L selects the per-label variant (single.cc): plain weight MPS, target y_n = [l_n == L];
        svd_backend = 1: the split on stock rocsolver_dsyevd (TNML_SVD_ROCSOLVER) instead of the in-house eigensolver"""
        self._L = _lib.load()
        self._h = C.c_void_p()
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        self.NT = int(labels.shape[0])
        self.N = int(N)
        self.single = single_label is not None
        self.c0 = -1 if self.single else self.N // 2
        self.nl = 1 if self.single else NL
        self.maxm = int(maxm)
        self.rank, self.nranks = rank, nranks
        self.NT_total = int(NT_total if NT_total is not None else self.NT)
        self.dtype = dtype
        cfg = _lib.Config(device, rank, nranks, self.N, self.NT, self.NT_total, self.maxm, _lib.DTYPES[dtype], int(svd_backend),
                          1 if self.single else 0, int(single_label) if self.single else 0)
        self._cfg = cfg
        rc = self._L.tnml_create(C.byref(self._h), C.byref(cfg))
        if rc != 0:
            self._h = C.c_void_p()
            raise TnmlError(self._L.tnml_last_error(None).decode())
        if pixels is not None:
            px = np.ascontiguousarray(pixels, dtype=np.uint8)
            assert px.shape == (self.NT, self.N)
            self._ck(self._L.tnml_set_data_u8(self._h, px.ctypes.data_as(C.POINTER(C.c_uint8)),
                                              labels.ctypes.data_as(C.POINTER(C.c_int32))))
        elif phi is not None:
            ph = np.ascontiguousarray(phi, dtype=np.float64)
            assert ph.shape == (self.NT, self.N, 2)
            self._ck(self._L.tnml_set_data_phi(self._h, _lib.dptr(ph), labels.ctypes.data_as(C.POINTER(C.c_int32))))
        else:
            raise ValueError("need pixels or phi")

    # -- plumbing
    def _ck(self, rc):
        if rc != 0:
            raise TnmlError(self._L.tnml_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.tnml_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_init(self, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        self._ck(self._L.tnml_comm_init(self._h, buf))

    @staticmethod
    def comm_init_local(states, oneshot=False):
        """in-process communicator: ranks sharing one device through a staging buffer (tnml_comm_init_local), or -- oneshot --
        the peer-write all-reduce for the ranks of one process on any devices (tnml_comm_init_oneshot); afterwards drive every
        rank from its own host thread"""
        arr = (C.c_void_p * len(states))(*[s._h for s in states])
        f = _lib.load().tnml_comm_init_oneshot if oneshot else _lib.load().tnml_comm_init_local
        if f(arr, len(states)) != 0:
            raise TnmlError(_lib.load().tnml_last_error(None).decode() or "in-process communicator setup failed")

    def oneshot_export(self) -> bytes:
        """the cross-process one-shot all-reduce, step 1 (tnml_oneshot_export): allocates this rank's receive region and returns the
        64-byte IPC handle the other ranks need"""
        buf = C.create_string_buffer(64)
        self._ck(self._L.tnml_oneshot_export(self._h, buf))
        return buf.raw

    def oneshot_connect(self, handles):
        """step 2 (tnml_oneshot_connect): `handles` = the handles of ALL ranks in rank order (gathered over any control plane)"""
        blob = b"".join(handles)
        assert len(blob) == 64 * self.nranks, (len(blob), self.nranks)
        self._ck(self._L.tnml_oneshot_connect(self._h, C.create_string_buffer(blob, len(blob))))

    def oneshot_region_bytes(self):
        """bytes of the receive region oneshot_export allocates (not part of estimate_bytes)"""
        return int(self._L.tnml_oneshot_region_bytes(C.byref(self._cfg)))

    def oneshot_mem_kind(self):
        """memory kind of the receive region of the cross-process one-shot transport: 1 fine-grained, 2 uncached, 0 none"""
        return self._L.tnml_oneshot_mem_kind(self._h)

    def collective_mode(self):
        """0 none (one rank), 1 RCCL, 2 in-process staging buffer, 3 in-process one-shot peer write, 4 cross-process one-shot (IPC)"""
        return self._L.tnml_collective_mode(self._h)

    def allreduce_mode(self):
        return ("none", "rccl", "in-process staging buffer", "one-shot peer write (threads of one process)",
                "one-shot peer write (processes, IPC-mapped receive regions, device-side arrival flags)")[self.collective_mode()]

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        if _lib.load().tnml_comm_unique_id(buf) != 0:
            raise TnmlError("tnml_comm_unique_id failed")
        return buf.raw

    def replica_check(self):
        """collective: communicator size == nranks and bit-identical W replicas on every rank; returns the communicator size"""
        n = C.c_int()
        self._ck(self._L.tnml_replica_check(self._h, C.byref(n)))
        return n.value

    def replica_repairs(self):
        return self._L.tnml_replica_repairs(self._h)

    def collective_stats(self):
        """(sum all-reduces, broadcasts) this rank has entered so far"""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self._L.tnml_collective_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_option(self, name, value):
        self._ck(self._L.tnml_set_option(self._h, name.encode(), int(value)))

    def size(self):
        return self.NT

    def synchronize(self):
        self._ck(self._L.tnml_synchronize(self._h))

    def svd_stats(self):
        fb, cr, d0, d1 = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        self._ck(self._L.tnml_svd_stats(self._h, C.byref(fb), C.byref(cr), C.byref(d0), C.byref(d1)))
        return dict(fallbacks=fb.value, cluster_repairs=cr.value, dev_before_polish=d0.value, dev_after_first_polish=d1.value)

    def split_stats(self):
        """speculative splits so far, how many tnml_bond_update_end rolled back (failed deferred check), device ms of the repeated work"""
        a, b, ms = C.c_int64(), C.c_int64(), C.c_double()
        self._ck(self._L.tnml_split_stats(self._h, C.byref(a), C.byref(b), C.byref(ms)))
        return dict(speculative_splits=a.value, roll_backs=b.value, roll_back_ms=ms.value)

    def device_bytes(self):
        return self._L.tnml_device_bytes(self._h)

    def estimate_bytes(self):
        """tnml_estimate_bytes for this context's configuration: what the drivers plan maxm with (tnml_plan_maxm)"""
        return int(self._L.tnml_estimate_bytes(C.byref(self._cfg)))

    def classify(self):
        """toverlap / fullTest (util.h:19-40,123-200) over the local images: returns (weights[NT,10], pred[NT],
        count[10], nincorrect[10])."""
        w = np.zeros((self.NT, self.nl))
        pred = np.zeros(self.NT, dtype=np.int32)
        cnt = np.zeros(10, dtype=np.int64)
        ninc = np.zeros(10, dtype=np.int64)
        self._ck(self._L.tnml_classify(self._h, _lib.dptr(w), pred.ctypes.data_as(C.POINTER(C.c_int32)),
                                       cnt.ctypes.data_as(C.POINTER(C.c_int64)), ninc.ctypes.data_as(C.POINTER(C.c_int64))))
        return w, pred, cnt, ninc

    # -- W
    def set_mps(self, W):
        for j, A in enumerate(W, start=1):
            A = np.asarray(A, dtype=np.float64)
            self._ck(self._L.tnml_set_site(self._h, j, A.shape[0], A.shape[2], int(A.ndim == 4), _lib.dptr(_lib.flat(A))))

    def set_site(self, j, A):
        A = np.asarray(A, dtype=np.float64)
        self._ck(self._L.tnml_set_site(self._h, j, A.shape[0], A.shape[2], int(A.ndim == 4), _lib.dptr(_lib.flat(A))))

    def get_site(self, j):
        ml, mr, hl = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.tnml_site_dims(self._h, j, ml, mr, hl))
        shape = (ml.value, 2, mr.value) + ((NL,) if hl.value else ())
        buf = np.empty(int(np.prod(shape)))
        self._ck(self._L.tnml_get_site(self._h, j, _lib.dptr(buf)))
        return buf.reshape(shape, order="F")

    def get_mps(self):
        return [self.get_site(j) for j in range(1, self.N + 1)]

    # -- TrainStates
    def init(self):
        self._ck(self._L.tnml_env_init(self._h))

    def setBond(self, b):
        self._ck(self._L.tnml_set_bond(self._h, b))

    def shiftE(self, b, from_left):
        self._ck(self._L.tnml_shift_env(self._h, b, int(bool(from_left))))

    def env_stats(self):
        """host tier of the environments (option env_budget_mb): copies to the host / back, slabs on the device, bytes on the host"""
        v = [C.c_int64() for _ in range(4)]
        self._ck(self._L.tnml_env_stats(self._h, *[C.byref(x) for x in v]))
        return dict(spills=v[0].value, fetches=v[1].value, slabs=v[2].value, host_bytes=v[3].value)

    def env(self, j):
        m, hl = C.c_int(), C.c_int()
        self._ck(self._L.tnml_env_dims(self._h, j, m, hl))
        L = NL if hl.value else 1
        buf = np.empty((self.NT, m.value * L))
        self._ck(self._L.tnml_get_env(self._h, j, _lib.dptr(buf)))
        if hl.value:
            return buf.reshape(self.NT, NL, m.value).transpose(0, 2, 1).copy()
        return buf

    # -- bond tensor and per-image contractions
    def bond_shape(self, b):
        mL, mR, lab = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.tnml_bond_dims(self._h, b, mL, mR, lab))
        return (mL.value, 2, 2, mR.value) + ((NL,) if lab.value else ())

    def bond_tensor(self, b):
        shape = self.bond_shape(b)
        buf = np.empty(int(np.prod(shape)))
        self._ck(self._L.tnml_bond_tensor(self._h, b, _lib.dptr(buf)))
        return buf.reshape(shape, order="F")

    def forward(self, B):
        P = np.empty((self.NT, self.nl))
        self._ck(self._L.tnml_forward(self._h, _lib.dptr(_lib.flat(B)), _lib.dptr(P)))
        return P[:, 0] if self.single else P

    def gradient(self, B):
        G = np.empty(B.size)
        self._ck(self._L.tnml_gradient(self._h, _lib.dptr(_lib.flat(B)), _lib.dptr(G)))
        return G.reshape(B.shape, order="F")

    def quadcost(self, B, lam):
        lc = np.empty(NL)
        cost, cr, nc = C.c_double(), C.c_double(), C.c_int64()
        self._ck(self._L.tnml_quadcost(self._h, _lib.dptr(_lib.flat(B)), lam, C.byref(cost), _lib.dptr(lc),
                                       C.byref(cr), C.byref(nc)))
        return cost.value, lc, cr.value, nc.value

    def pAp(self, p, lam):
        """sum_n |p*t.v_n|^2 + lambda |p|^2 (fixedL.cc:394-403)"""
        out = C.c_double()
        self._ck(self._L.tnml_pAp(self._h, _lib.dptr(_lib.flat(p)), lam, C.byref(out)))
        return out.value

    def cgrad(self, B, npass, lam, cconv):
        buf = _lib.flat(B)
        tr = _lib.CgTrace()
        self._ck(self._L.tnml_cgrad(self._h, _lib.dptr(buf), npass, lam, cconv, C.byref(tr)))
        return buf.reshape(B.shape, order="F"), _trace_dict(tr)

    def exact(self, b, lam, pcut=1e-8):
        """single.h:117-160 on the bond chosen by setBond(b) (per-label variant): the solved bond tensor"""
        shape = self.bond_tensor(b).shape
        buf = np.zeros(int(np.prod(shape)))
        self._ck(self._L.tnml_exact(self._h, _lib.dptr(buf), lam, pcut))
        return buf.reshape(shape, order="F")

    def pinv(self, b, V0, npass, lam, pcut=1e-8):
        """single.h:404-517 on the bond chosen by setBond(b) from the start V0 [D, r]: (B, trace of V*E, singular values of the last E)"""
        shape = self.bond_tensor(b).shape
        V0 = np.asfortranarray(V0, dtype=np.float64)
        D, r = V0.shape
        assert D == int(np.prod(shape))
        B = np.zeros(D); ve = np.zeros(npass + 1); Dsv = np.zeros(r); done = C.c_int()
        self._ck(self._L.tnml_pinv(self._h, _lib.dptr(V0), r, npass, lam, pcut, _lib.dptr(B), _lib.dptr(ve), C.byref(done), _lib.dptr(Dsv)))
        return B.reshape(shape, order="F"), ve[:done.value + 1].copy(), Dsv

    def set_option_real(self, name, value):
        self._ck(self._L.tnml_set_option_real(self._h, name.encode(), float(value)))

    def svd_split(self, B, b, ha, cutoff, maxm, minm):
        te, m, nsv = C.c_double(), C.c_int(), C.c_int()
        sv = np.empty(4 * self.maxm + 8)
        self._ck(self._L.tnml_svd_split(self._h, _lib.dptr(_lib.flat(B)), b, ha, cutoff, maxm, minm, C.byref(te),
                                        C.byref(m), _lib.dptr(sv), C.byref(nsv)))
        return m.value, te.value, sv[:nsv.value].copy()

    def bond_update(self, b, ha, maxm, minm, cutoff, npass, lam, cconv, lam_cost=None, report_costs=False):
        sp = _lib.SweepParams(maxm, minm, cutoff, npass, lam, lam if lam_cost is None else lam_cost, cconv, int(report_costs))
        rep = _lib.BondReport()
        self._ck(self._L.tnml_bond_update(self._h, b, ha, C.byref(sp), C.byref(rep)))
        return self._report(rep)

    def bond_update_begin(self, b, ha, maxm, minm, cutoff, npass, lam, cconv, lam_cost=None, report_costs=False):
        """enqueue one bond update; the report comes from bond_update_end (at most two bond updates in flight)"""
        sp = _lib.SweepParams(maxm, minm, cutoff, npass, lam, lam if lam_cost is None else lam_cost, cconv, int(report_costs))
        self._ck(self._L.tnml_bond_update_begin(self._h, b, ha, C.byref(sp)))

    def bond_update_end(self):
        rep = _lib.BondReport()
        self._ck(self._L.tnml_bond_update_end(self._h, C.byref(rep)))
        return self._report(rep)

    @staticmethod
    def _report(rep):
        return dict(bond=rep.bond, half=rep.half, c=rep.c, mL=rep.mL, mR=rep.mR, label_on_B=bool(rep.label_on_B), origm=rep.origm, newm=rep.newm, truncerr=rep.truncerr,
                    norm_newB=rep.norm_newB, diff=rep.diff_B_newB, cost=rep.cost_after_svd,
                    label_cost=np.array(rep.label_cost[:]), reg_cost=rep.reg_cost, ncorrect=rep.ncorrect,
                    cg=_trace_dict(rep.cg), cost_old=rep.cost_old, cost_cg=rep.cost_cg, reg_cost_cg=rep.reg_cost_cg, norm_oB=rep.norm_oB)

    # -- measurement
    def profile(self, on, only=None):
        """HIP-event timing of kernel launches; `only` restricts it to one kernel class"""
        self._ck(self._L.tnml_profile_select(self._h, (only or "").encode()))
        self._ck(self._L.tnml_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._ck(self._L.tnml_profile_reset(self._h))

    def profile_read(self):
        out = {}
        name = C.create_string_buffer(64)
        for i in range(self._L.tnml_profile_count(self._h)):
            n, ms = C.c_int64(), C.c_double()
            self._ck(self._L.tnml_profile_get(self._h, i, name, C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (n.value, ms.value)
        return out


def _trace_dict(tr):
    n = tr.npass_done
    k = n if tr.converged else max(n - 1, 0)
    return dict(npass_done=n, converged=bool(tr.converged), skipped=tr.converged == 2, cost=list(tr.cost[:k]), rnorm=list(tr.rnorm[:k]),
                pAp=list(tr.pAp[:n]), alpha=list(tr.alpha[:n]))


def quadcost(B, ts, lam=0.0):
    """fixedL.cc:280-344: returns the un-normalised cost C = sum_l C_l + lambda|B|^2"""
    return ts.quadcost(B, lam)[0]


def cgrad(B, ts, npass=4, lam=0.0, cconv=1e-10):
    """fixedL.cc:349-445"""
    return ts.cgrad(B, npass, lam, cconv)


def mldmrg(ts, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds=0, log=None, report_costs=False, pipelined=False):
    """fixedL.cc:451-570 (single.h:523-728 for a per-label TrainStates): the sweep loop; emits the reference's log lines
    through `log` if given.  pipelined: bond k+1 is enqueued before the report of bond k is fetched (same results, no idle
    GPU between bond updates; the log of a bond appears one bond later)."""
    NT = float(ts.NT_total)
    reports = []
    if pipelined and not log:
        sched = []
        for sw in range(1, nsweep + 1):
            b, ha = 1, 1
            while ha <= 2:
                sched.append((sw, b, ha))
                b, ha = _lib.sweepnext(b, ha, ts.N)
        if max_bonds:
            sched = sched[:max_bonds]
        for k, (sw, b, ha) in enumerate(sched):
            ts.bond_update_begin(b, ha, maxm, minm, cutoff, npass, lam, cconv, report_costs=report_costs or ts.single)
            if k > 0:
                r = ts.bond_update_end()
                r["sweep"] = sched[k - 1][0]
                reports.append(r)
        if sched:
            r = ts.bond_update_end()
            r["sweep"] = sched[-1][0]
            reports.append(r)
        return reports
    for sw in range(1, nsweep + 1):
        if log:
            log("\nSweep %d maxm=%d minm=%d" % (sw, maxm, minm))
        b, ha = 1, 1
        while ha <= 2:
            if max_bonds and len(reports) >= max_bonds:
                return reports
            r = ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, cconv, report_costs=report_costs or ts.single)
            r["sweep"] = sw
            reports.append(r)
            if log:
                log("Sweep %d Half %d Bond %d" % (sw, ha, r["c"]))
                log("In cgrad, lambda = %.3E" % lam)
                for p in range(r["cg"]["npass_done"]):
                    log("  Conj grad pass %d" % (p + 1))
                    if p < len(r["cg"]["cost"]):
                        log("  Cost = %.10f" % (r["cg"]["cost"][p] / NT))
                        log("  |r| = %.1E" % r["cg"]["rnorm"][p])
                log("SVD trunc err = %.2E" % r["truncerr"])
                log("Original m=%d, New m=%d" % (r["origm"], r["newm"]))
                log("|B-newB| = %.3E" % r["diff"])
                for l in range(NL):
                    log("  Label l=%d C%d = %.10f" % (l, l, r["label_cost"][l] / NT))
                log("  Reg. cost CR = %.10f" % (r["reg_cost"] / NT))
                log("Percent correct = %.4f%%, # incorrect = %d/%d" % (r["ncorrect"] * 100.0 / NT, NT - r["ncorrect"], NT))
                log("--> After SVD, Cost = %.10f" % (r["cost"] / NT))
            b, ha = _lib.sweepnext(b, ha, ts.N)
    return reports
