"""Independent numpy restatement of the fixedL hot path (TEST INFRASTRUCTURE ONLY).

Second, independently written restatement of /root/reference/fixedL.cc used to cross-check the
C oracle (oracle/fixedl_oracle.c) on tiny shapes: einsum contractions by index name instead of
hand-written loops, numpy.linalg.svd instead of the Jacobi SVD.  PARITY UNPINNED: see the C
oracle's header -- the reference has no tests/golden vectors and ITensor is unavailable.

Arrays carry ITensor index order as numpy axes: A_j[a,s,r(,L)], B[a,s,t,r(,L)], E[n,m(,L)].
Sites/bonds are 1-indexed as in the reference; lists are padded at index 0.
"""
import numpy as np

NL = 10


def features_series(pixels):
    """fixedL.cc:637-642 after mllib/mnist.h:495: phi(g,n) = ((g/255)/4)**(n-1), g = byte/255."""
    x = pixels.astype(np.float64) / 255.0 / 255.0
    return np.stack([np.ones_like(x), x / 4.0], axis=-1)          # [NT,N,2]


def truncate(p, maxm, minm, cutoff):
    """ITensor truncate() as recalled in SURVEY.md 8(a9); p descending. Returns (m, truncerr)."""
    origm = len(p)
    if origm == 1:
        return 1, 0.0
    n = origm - 1
    te = 0.0
    while n >= maxm:
        te += p[n]
        n -= 1
    scale = float(np.sum(p)) or 1.0
    while n >= 0 and te + p[n] < cutoff * scale and n >= minm:
        te += p[n]
        n -= 1
    n = max(n, 0)
    return n + 1, te / scale


def sweepnext(b, ha, N):
    inc = 1 if ha == 1 else -1
    b += inc
    if b == (N if ha == 1 else 0):
        b -= inc
        ha += 1
    return b, ha


class NpFixedL:
    def __init__(self, phi, labels, W):
        self.phi = np.asarray(phi, dtype=np.float64)             # [NT,N,2]
        self.labels = np.asarray(labels)
        self.NT, self.N, _ = self.phi.shape
        self.c0 = self.N // 2
        self.W = [None] + [np.array(a, dtype=np.float64) for a in W]    # 1..N
        self.E = [None] * (self.N + 2)
        self.v = None
        self.currb = -1
        self.delta = np.eye(NL)[self.labels]                      # [NT,10]

    # --- fixedL.cc:122-157
    def init(self):
        N = self.N
        for n in range(N, 2, -1):
            M = np.einsum('ns,asr...->nar...', self.phi[:, n - 1], self.W[n])   # t.A(n)*W.A(n)
            if n == N:
                self.E[n] = M[:, :, 0]
            elif M.ndim == 4:                                     # label on this site
                self.E[n] = np.einsum('narl,nr->nal', M, self.E[n + 1])
            elif self.E[n + 1].ndim == 3:
                self.E[n] = np.einsum('nar,nrl->nal', M, self.E[n + 1])
            else:
                self.E[n] = np.einsum('nar,nr->na', M, self.E[n + 1])
        self.currb = -1
        self.set_bond(1)

    # --- fixedL.cc:159-190
    def set_bond(self, b):
        if self.currb == b:
            return
        self.currb = b
        lc, rc = b - 1, b + 2
        v = np.einsum('ns,nt->nst', self.phi[:, b - 1], self.phi[:, b])
        LE = self.E[lc] if lc > 0 else np.ones((self.NT, 1))
        RE = self.E[rc] if rc < self.N + 1 else np.ones((self.NT, 1))
        if LE.ndim == 3:
            self.v = np.einsum('nal,nst,nr->nastrl', LE, v, RE)
        elif RE.ndim == 3:
            self.v = np.einsum('na,nst,nrl->nastrl', LE, v, RE)
        else:
            self.v = np.einsum('na,nst,nr->nastr', LE, v, RE)

    # --- fixedL.cc:192-233
    def shiftE(self, b, from_left):
        c = b if from_left else b + 1
        prevc = b - 1 if from_left else b + 2
        M = np.einsum('ns,asr...->nar...', self.phi[:, c - 1], self.W[c])
        if not (1 <= prevc <= self.N):
            E = M[:, 0] if from_left else M[:, :, 0]
        else:
            P = self.E[prevc]
            lab_m, lab_p = M.ndim == 4, P.ndim == 3
            if from_left:
                sub = ('narl,na->nrl' if lab_m else 'nar,nal->nrl' if lab_p else 'nar,na->nr')
            else:
                sub = ('narl,nr->nal' if lab_m else 'nar,nrl->nal' if lab_p else 'nar,nr->na')
            E = np.einsum(sub, M, P)
        self.E[c] = E

    def bond_tensor(self, b):
        A1, A2 = self.W[b], self.W[b + 1]
        if A1.ndim == 4:
            return np.einsum('asgl,gtr->astrl', A1, A2)
        if A2.ndim == 4:
            return np.einsum('asg,gtrl->astrl', A1, A2)
        return np.einsum('asg,gtr->astr', A1, A2)

    # --- B*t.v, dP*dag(t.v)
    def forward(self, B):
        if self.v.ndim == 6:
            return np.einsum('astr,nastrl->nl', B, self.v)
        return np.einsum('astrl,nastr->nl', B, self.v)

    def gradient(self, B):
        dP = self.delta - self.forward(B)
        return self.backward(dP)

    def backward(self, dP):
        if self.v.ndim == 6:
            return np.einsum('nl,nastrl->astr', dP, self.v)
        return np.einsum('nl,nastr->astrl', dP, self.v)

    # --- fixedL.cc:280-344
    def quadcost(self, B, lam):
        P = self.forward(B)
        dP = self.delta - P
        per_img = np.sum(dP ** 2, axis=1)
        label_cost = np.array([per_img[self.labels == l].sum() for l in range(NL)])
        pred = np.argmax(np.abs(P), axis=1)                       # first maximum, util.h:42-57
        ncor = int(np.sum(pred == self.labels))
        CR = lam * np.sum(B ** 2)
        return label_cost.sum() + CR, label_cost, CR, ncor

    # --- fixedL.cc:349-445
    def cgrad(self, B, npass, lam, cconv):
        B = B.copy()
        trace = {'cost': [], 'rnorm': [], 'pAp': [], 'alpha': [], 'converged': False}
        r = self.gradient(B)
        if lam != 0.0:
            r = r - lam * B
        p = r.copy()
        for ps in range(1, npass + 1):
            pv = self.forward(p)
            pAp = np.sum(pv ** 2) + lam * np.sum(p ** 2)
            a = np.sum(r ** 2) / pAp
            B = B + a * p
            trace['pAp'].append(pAp)
            trace['alpha'].append(a)
            if ps == npass:
                break
            dP = self.delta - self.forward(B)
            nr = self.backward(dP)
            if lam != 0.0:
                nr = nr - lam * B
            beta = (np.linalg.norm(nr) / np.linalg.norm(r)) ** 2
            r = nr
            C = np.sum(dP ** 2) + lam * np.sum(B ** 2)
            trace['cost'].append(C)
            trace['rnorm'].append(np.linalg.norm(r))
            if np.linalg.norm(r) < cconv:
                trace['converged'] = True
                break
            p = r + beta * p
        return B, trace

    # --- fixedL.cc:519-521
    def svd_split(self, B, b, ha, cutoff, maxm, minm):
        labL, labR = self.c0 == b, self.c0 == b + 1
        mL, mR = B.shape[0], B.shape[3]
        if labL:
            M = np.transpose(B, (0, 1, 4, 2, 3)).reshape(mL * 2 * NL, 2 * mR)       # rows (a,s,l)
        elif labR:
            M = B.reshape(mL * 2, 2 * mR * NL)                                       # cols (t,r,l)
        else:
            M = B.reshape(mL * 2, 2 * mR)
        U, s, Vt = np.linalg.svd(M, full_matrices=False)
        m, te = truncate(s ** 2, maxm, minm, cutoff)
        U, s_k, Vt = U[:, :m], s[:m], Vt[:m]
        if ha == 1:
            left, right = U, s_k[:, None] * Vt
        else:
            left, right = U * s_k[None, :], Vt
        if labL:
            A1 = np.transpose(left.reshape(mL, 2, NL, m), (0, 1, 3, 2))
        else:
            A1 = left.reshape(mL, 2, m)
        if labR:
            A2 = right.reshape(m, 2, mR, NL)
        else:
            A2 = right.reshape(m, 2, mR)
        self.W[b], self.W[b + 1] = A1, A2
        return m, te, s

    # --- fixedL.cc:451-570
    def mldmrg(self, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds=0):
        reports = []
        for sw in range(1, nsweep + 1):
            b, ha = 1, 1
            while ha <= 2:
                if max_bonds and len(reports) >= max_bonds:
                    return reports
                self.set_bond(b)
                oB = self.bond_tensor(b)
                origm = self.W[b].shape[2]
                B, tr = self.cgrad(oB, npass, lam, cconv)
                m, te, s = self.svd_split(B, b, ha, cutoff, maxm, minm)
                newB = self.bond_tensor(b)
                C, lc, CR, ncor = self.quadcost(newB, lam)
                self.shiftE(b, ha == 1)
                reports.append(dict(sweep=sw, half=ha, bond=b, origm=origm, newm=m, truncerr=te,
                                    diff=np.linalg.norm(B - newB), cost=C, label_cost=lc, ncorrect=ncor,
                                    cg=tr, sv=s))
                b, ha = sweepnext(b, ha, self.N)
        return reports

    # --- util.h:19-40
    def toverlap(self, i):
        N, c = self.N, self.c0
        ph = self.phi[i]
        cur = np.einsum('s,asr...->ar...', ph[N - 1], self.W[N])[:, 0]
        for j in range(N - 1, c - 1, -1):
            M = np.einsum('s,asr...->ar...', ph[j - 1], self.W[j])
            if M.ndim == 3:
                cur = np.einsum('arl,r->al', M, cur)
            elif cur.ndim == 2:
                cur = np.einsum('ar,rl->al', M, cur)
            else:
                cur = M @ cur
        left = np.einsum('s,asr->ar', ph[0], self.W[1])[0]
        for j in range(2, c):
            left = left @ np.einsum('s,asr->ar', ph[j - 1], self.W[j])
        return left @ cur


# ---------------------------------------------------------------------------------------------------
# per-label variant: /root/reference/single.cc, single.h (plain MPS, scalar output regressed on [l == L])
def features_single(pixels, normal=True):
    """single.cc:71-84 with g = byte/255 (mllib/mnist.h:495): x = g/255"""
    x = pixels.astype(np.float64) / 255.0 / 255.0
    if normal:
        return np.stack([np.cos(np.pi / 2 * x), np.sin(np.pi / 2 * x)], axis=-1)
    return np.stack([np.ones_like(x), x / 4.0], axis=-1)


class NpSingle:
    def __init__(self, phi, labels, target, W):
        self.phi = np.asarray(phi, dtype=np.float64)
        self.NT, self.N, _ = self.phi.shape
        self.y = (np.asarray(labels) == target).astype(np.float64)          # single.h:103,193
        self.W = [None] + [np.array(a, dtype=np.float64) for a in W]
        self.E = [None] * (self.N + 2)
        self.v = None

    def init(self):                                                          # single.cc:181-199
        N = self.N
        for n in range(N, 2, -1):
            M = np.einsum('ns,asr->nar', self.phi[:, n - 1], self.W[n])
            self.E[n] = M[:, :, 0] if n == N else np.einsum('nar,nr->na', M, self.E[n + 1])
        self.set_bond(1)

    def set_bond(self, b):                                                   # single.h:581-596
        p1, p2 = self.phi[:, b - 1], self.phi[:, b]
        LE = self.E[b - 1] if b - 1 > 0 else np.ones((self.NT, 1))
        RE = self.E[b + 2] if b + 2 < self.N + 1 else np.ones((self.NT, 1))
        self.v = np.einsum('na,ns,nt,nr->nastr', LE, p1, p2, RE)

    def shiftE(self, b, from_left):                                          # single.h:688-710
        c, dc = (b, 1) if from_left else (b + 1, -1)
        M = np.einsum('ns,asr->nar', self.phi[:, c - 1], self.W[c])
        if c == 1 or c == self.N:
            self.E[c] = M[:, 0, :] if from_left else M[:, :, 0]
        elif from_left:
            self.E[c] = np.einsum('na,nar->nr', self.E[c - dc], M)
        else:
            self.E[c] = np.einsum('nar,nr->na', M, self.E[c - dc])

    def bond_tensor(self, b):
        return np.einsum('asg,gtr->astr', self.W[b], self.W[b + 1])

    def forward(self, B):
        return np.einsum('astr,nastr->n', B, self.v)

    def gradient(self, B):
        return np.einsum('n,nastr->astr', self.y - self.forward(B), self.v)

    def quadcost(self, B, lam):
        return float(np.sum((self.y - self.forward(B)) ** 2) + lam * np.sum(B * B))

    def cgrad(self, B, npass, lam, cconv):                                   # single.h:162-288
        B = B.copy()
        r = self.gradient(B) - lam * B
        trace = dict(skipped=False, cost=[], rnorm=[], alpha=[])
        if np.linalg.norm(r) < cconv:
            trace["skipped"] = True
            return B, trace
        p = r.copy()
        for ps in range(1, npass + 1):
            pAp = float(np.sum(self.forward(p) ** 2) + lam * np.sum(p * p))
            a = float(np.sum(r * r)) / pAp
            trace["alpha"].append(a)
            B = B + a * p
            if ps == npass:
                break
            nr = self.gradient(B) - lam * B
            beta = (np.linalg.norm(nr) / np.linalg.norm(r)) ** 2
            r = nr
            trace["cost"].append(self.quadcost(B, lam))
            trace["rnorm"].append(float(np.linalg.norm(r)))
            if np.linalg.norm(r) < cconv:
                break
            p = r + beta * p
        return B, trace

    def exact(self, lam, pcut=1e-8):                                         # single.h:117-160
        """B = y Phi^+ with the filtered inverse s/(s^2 + lambda) above pcut (numpy.linalg.svd of the NT x D matrix of the v_n)"""
        shp = self.v.shape[1:]
        Phi = self.v.reshape(self.v.shape[0], -1)                            # [n][D], C order of (a, s, t, r)
        U, sv, Vt = np.linalg.svd(Phi, full_matrices=False)
        f = np.where(sv > pcut, sv / (sv * sv + lam), 0.0)
        return ((self.y @ U) * f @ Vt).reshape(shp)

    def pinv(self, V0, npass, lam, pcut=1e-8):                               # single.h:404-517
        """subspace iteration on A = Phi^T Phi from the start V0 [D, r] (D in Fortran order of (a, s, t, r)), then B = yUS Einv"""
        shp = self.v.shape[1:]
        Phi = np.stack([x.reshape(-1, order="F") for x in self.v])          # [n][D], D column-major like the C oracle's tensors
        Uv, _, Wv = np.linalg.svd(V0, full_matrices=False)
        V = Uv @ Wv                                                          # polarU
        E = (Phi @ V).T @ Phi                                                # r x D
        last = float(np.sum(V.T * E))
        ve = [last]
        F = Dg = G = None
        for _ in range(npass):
            E = (Phi @ V).T @ Phi
            F, Dg, G = np.linalg.svd(E, full_matrices=False)                 # E = F diag(D) G
            V = (F @ G).T
            VE = float(np.sum(V.T * E))
            ve.append(VE)
            if abs(VE - last) < 1e-4:
                break
            last = VE
        if F is None:
            return np.zeros(shp), np.array(ve), None
        f = np.where(Dg > pcut, Dg / (Dg * Dg + lam), 0.0)
        yus = (self.y @ Phi) @ V
        B = ((yus @ F) * f) @ G
        return B.reshape(shp, order="F"), np.array(ve), Dg

    def fast_cgrad(self, B, npass, lam, cconv):                              # single.h:290-398
        """one contraction with the images per step: p.v gives pAp and A p; residual by recurrence, with the reference's
        'nr = nr - lambda*B' (:379) as written"""
        B = B.copy()
        r = self.gradient(B) - lam * B
        trace = dict(skipped=False, cost=[], rnorm=[], alpha=[])
        if np.linalg.norm(r) < cconv:
            trace["skipped"] = True
            return B, trace
        p = r.copy()
        for ps in range(1, npass + 1):
            pv = self.forward(p)
            pAp = float(np.sum(pv ** 2) + lam * np.sum(p * p))
            a = float(np.sum(r * r)) / pAp
            trace["alpha"].append(a)
            B = B + a * p
            if ps == npass:
                break
            Ap = np.einsum('n,nastr->astr', pv, self.v)
            nr = r - a * Ap
            if lam != 0.:
                nr = nr - lam * B
            beta = (np.linalg.norm(nr) / np.linalg.norm(r)) ** 2
            r = nr
            trace["rnorm"].append(float(np.linalg.norm(r)))
            if np.linalg.norm(r) < cconv:
                break
            p = r + beta * p
        return B, trace

    def svd_split(self, B, b, ha, cutoff, maxm, minm):                       # single.h:636-646
        mL, _, _, mR = B.shape
        M = B.reshape(2 * mL, 2 * mR, order="F") if False else np.einsum('astr->astr', B).reshape(mL * 2, 2 * mR)
        # rows (a,s), cols (t,r) in C order of the reshaped einsum view
        U, s, Vt = np.linalg.svd(M, full_matrices=False)
        m, te = truncate(s ** 2, maxm, minm, cutoff)
        U, s, Vt = U[:, :m], s[:m], Vt[:m]
        if ha == 1:
            self.W[b] = U.reshape(mL, 2, m)
            self.W[b + 1] = (s[:, None] * Vt).reshape(m, 2, mR)
        else:
            self.W[b] = (U * s).reshape(mL, 2, m)
            self.W[b + 1] = Vt.reshape(m, 2, mR)
        return m, te

    def noise_split(self, B, b, ha, noise, cutoff, maxm, minm):              # single.h:648-672
        """density-matrix split with a noise term, written on the tensors themselves: rho over the indices (link, site) of site c, the
        images' contribution dr_n = (B . E_n) (x) E_n contracted with itself over the indices of the other site"""
        mL, _, _, mR = B.shape
        c = b if ha == 1 else b + 1
        if ha == 1:                                                          # site c = b carries (a, s); the other site (t, r)
            rho = np.einsum('astr,butr->asbu', B, B)
            if c > 1:
                E = self.E[c - 1]
                T = np.einsum('na,astr->nstr', E, B)
                w = np.einsum('nstr,nutr->nsu', T, T)
                drho = np.einsum('nsu,na,nb->asbu', w, E, E)
            else:
                drho = self.NT * rho
            rho = (rho + noise * drho).reshape(2 * mL, 2 * mL)               # C order: row index (a, s)
        else:                                                                # site c = b + 1 carries (t, r); the other site (a, s)
            rho = np.einsum('astr,asuq->truq', B, B)
            if c < self.N - 1:                                              # single.h:655 as written: "ha == 2 && c < N-1"
                E = self.E[c + 1]
                T = np.einsum('nr,astr->nast', E, B)
                w = np.einsum('nast,nasu->ntu', T, T)
                drho = np.einsum('ntu,nr,nq->truq', w, E, E)
            else:
                drho = self.NT * rho
            rho = (rho + noise * drho).reshape(2 * mR, 2 * mR)               # row index (t, r)
        ev, U = np.linalg.eigh(rho)
        ev, U = ev[::-1], U[:, ::-1]
        m, te = truncate(np.maximum(ev, 0.), maxm, minm, cutoff)
        U = U[:, :m]
        if ha == 1:
            self.W[b] = U.reshape(mL, 2, m)
            self.W[b + 1] = np.einsum('asg,astr->gtr', self.W[b], B)
        else:
            self.W[b + 1] = U.reshape(2, mR, m).transpose(2, 0, 1)
            self.W[b] = np.einsum('gtr,astr->asg', self.W[b + 1], B)
        return m, te

    def mldmrg(self, nsweep, maxm, minm, cutoff, npass, lam, cconv, max_bonds=0):
        out = []
        for sw in range(1, nsweep + 1):
            b, ha = 1, 1
            while ha <= 2:
                if max_bonds and len(out) >= max_bonds:
                    return out
                oB = self.bond_tensor(b)
                self.set_bond(b)
                B, tr = self.cgrad(oB, npass, lam, cconv)
                rep = dict(c=b if ha == 1 else b + 1, half=ha, origm=self.W[b].shape[2], cost_old=self.quadcost(oB, lam),
                           cost_cg=self.quadcost(B, lam), cg_skipped=tr["skipped"])
                if getattr(self, "noise", 0.) < 1e-14:
                    rep["newm"], rep["truncerr"] = self.svd_split(B, b, ha, cutoff, maxm, minm)
                else:
                    rep["newm"], rep["truncerr"] = self.noise_split(B, b, ha, self.noise, cutoff, maxm, minm)
                rep["cost"] = self.quadcost(self.bond_tensor(b), lam)
                self.shiftE(b, ha == 1)
                out.append(rep)
                b, ha = sweepnext(b, ha, self.N)
        return out

    def output(self, i):
        cur = np.ones(1)
        for j in range(self.N, 0, -1):
            cur = np.einsum('s,asr,r->a', self.phi[i, j - 1], self.W[j], cur)
        return float(cur[0])
