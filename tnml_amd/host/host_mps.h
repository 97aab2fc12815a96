// host_mps.h -- host-side weight MPS: container, on-disk format, product states, sums with truncation.
//
// Replaces the parts of ITensor's MPS class the fixedL driver touches outside the hot path:
//   readFromFile/writeToFile("W")  fixedL.cc:674,727,764   -> own versioned format "TNMLW1" (the ITensor v2
//       binary dump is not reproducible offline); same file names and semantics
//   makeMPS(sites,img,phi)         util.h:76-102           -> product_state()
//   sum(psis,{"Cutoff",..,"Maxm",..}) fixedL.cc:697,720,724 -> add() + compress()
//   overlap(W,W)                   fixedL.cc:729           -> overlap()
// Tensor layout everywhere: A_j[l][s][r]([L]) column-major, Label (dim 10) only on site c0 = N/2.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace tnmlh {

constexpr int NL = 10;

struct Site {
    int ml = 1, mr = 1, L = 1;
    std::vector<double> a;                       // [ml][2][mr][L]
    Site() = default;
    Site(int ml_, int mr_, int L_) : ml(ml_), mr(mr_), L(L_), a((size_t)ml_ * 2 * mr_ * L_, 0.) {}
    double& at(int l, int s, int r, int lab = 0) { return a[l + (size_t)ml * (s + 2 * (r + (size_t)mr * lab))]; }
    double at(int l, int s, int r, int lab = 0) const { return a[l + (size_t)ml * (s + 2 * (r + (size_t)mr * lab))]; }
};

struct HostMPS {
    int N = 0, c0 = 0;
    std::vector<Site> A;                          // 1..N (A[0] unused)
    HostMPS() = default;
    explicit HostMPS(int N_) : N(N_), c0(N_ / 2), A(N_ + 1) {}
};

// ---- dense helpers ---------------------------------------------------------------------------
// thin SVD of the R x C column-major matrix M by one-sided Jacobi: M = U diag(s) Vt, k = min(R,C), s descending
inline void jacobi_svd(int R, int C, const std::vector<double>& M, std::vector<double>& U, std::vector<double>& s, std::vector<double>& Vt) {
    const bool tall = R >= C;
    const int r = tall ? R : C, c = tall ? C : R, k = c;
    std::vector<double> W((size_t)r * c), V((size_t)c * c, 0.);
    if (tall) W = M;
    else for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) W[j + (size_t)C * i] = M[i + (size_t)R * j];
    for (int j = 0; j < c; ++j) V[j + (size_t)c * j] = 1.;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rot = false;
        for (int p = 0; p < c - 1; ++p) for (int q = p + 1; q < c; ++q) {
            double* wp = &W[(size_t)r * p]; double* wq = &W[(size_t)r * q];
            double al = 0., be = 0., ga = 0.;
            for (int i = 0; i < r; ++i) { al += wp[i] * wp[i]; be += wq[i] * wq[i]; ga += wp[i] * wq[i]; }
            if (ga == 0. || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
            rot = true;
            const double zeta = (be - al) / (2. * ga);
            const double t = (zeta >= 0. ? 1. : -1.) / (std::fabs(zeta) + std::sqrt(1. + zeta * zeta));
            const double cs = 1. / std::sqrt(1. + t * t), sn = cs * t;
            for (int i = 0; i < r; ++i) { const double x = wp[i], y = wq[i]; wp[i] = cs * x - sn * y; wq[i] = sn * x + cs * y; }
            double* vp = &V[(size_t)c * p]; double* vq = &V[(size_t)c * q];
            for (int i = 0; i < c; ++i) { const double x = vp[i], y = vq[i]; vp[i] = cs * x - sn * y; vq[i] = sn * x + cs * y; }
        }
        if (!rot) break;
    }
    std::vector<double> sv(c);
    std::vector<int> ord(c);
    for (int j = 0; j < c; ++j) {
        double n2 = 0.; for (int i = 0; i < r; ++i) n2 += W[i + (size_t)r * j] * W[i + (size_t)r * j];
        sv[j] = std::sqrt(n2); ord[j] = j;
        if (sv[j] > 0.) for (int i = 0; i < r; ++i) W[i + (size_t)r * j] /= sv[j];
    }
    for (int i = 1; i < c; ++i) { int x = ord[i], j = i - 1; while (j >= 0 && sv[ord[j]] < sv[x]) { ord[j + 1] = ord[j]; --j; } ord[j + 1] = x; }
    U.assign((size_t)R * k, 0.); s.assign(k, 0.); Vt.assign((size_t)k * C, 0.);
    for (int g = 0; g < k; ++g) {
        const int j = ord[g];
        s[g] = sv[j];
        if (tall) { for (int i = 0; i < R; ++i) U[i + (size_t)R * g] = W[i + (size_t)r * j]; for (int i = 0; i < C; ++i) Vt[g + (size_t)k * i] = V[i + (size_t)c * j]; }
        else      { for (int i = 0; i < R; ++i) U[i + (size_t)R * g] = V[i + (size_t)c * j]; for (int i = 0; i < C; ++i) Vt[g + (size_t)k * i] = W[i + (size_t)r * j]; }
    }
}

// ITensor truncate() -- same rule as tnml_truncate
inline int truncate_rule(const std::vector<double>& P, int maxm, int minm, double cutoff, double* truncerr = nullptr) {
    const int origm = (int)P.size();
    if (origm <= 1) { if (truncerr) *truncerr = 0.; return origm; }
    int n = origm - 1; double te = 0.;
    while (n >= maxm) { te += P[n]; --n; }
    double scale = 0.; for (double p : P) scale += p;
    if (scale == 0.) scale = 1.;
    while (n >= 0 && te + P[n] < cutoff * scale && n >= minm) { te += P[n]; --n; }
    if (n < 0) n = 0;
    if (truncerr) *truncerr = te / scale;
    return n + 1;
}

// ---- construction ----------------------------------------------------------------------------
// makeMPS(sites,img,phi), util.h:76-102: bond dimension 1, A_j[0][s][0] = phi(pixel_j, s+1); the Label
// index is attached by the caller (scale_label)
inline HostMPS product_state(int N, const double* phi /* [N][2] */) {
    HostMPS psi(N);
    for (int j = 1; j <= N; ++j) { psi.A[j] = Site(1, 1, 1); psi.A[j].at(0, 0, 0) = phi[(j - 1) * 2]; psi.A[j].at(0, 1, 0) = phi[(j - 1) * 2 + 1]; }
    return psi;
}
// in.Aref(c) *= w*setElt(L(label))  (fixedL.cc:693,721): put the Label index on site c0, weight w on entry `label`
inline void attach_label(HostMPS& psi, int label, double w) {
    Site& s = psi.A[psi.c0];
    if (s.L != 1) throw std::runtime_error("attach_label: site already carries a Label index");
    Site t(s.ml, s.mr, NL);
    for (int r = 0; r < s.mr; ++r) for (int sg = 0; sg < 2; ++sg) for (int l = 0; l < s.ml; ++l) t.at(l, sg, r, label) = w * s.at(l, sg, r);
    s = t;
}
// direct sum psi + phi (block-diagonal links; the Label index, a physical index, is shared)
inline HostMPS add(const HostMPS& x, const HostMPS& y) {
    if (x.N != y.N) throw std::runtime_error("add: different lengths");
    HostMPS z(x.N);
    for (int j = 1; j <= x.N; ++j) {
        const Site &a = x.A[j], &b = y.A[j];
        if (a.L != b.L) throw std::runtime_error("add: Label index on different sites");
        const int ml = j == 1 ? 1 : a.ml + b.ml, mr = j == x.N ? 1 : a.mr + b.mr;
        Site c(ml, mr, a.L);
        for (int lab = 0; lab < a.L; ++lab) for (int s = 0; s < 2; ++s) {
            for (int r = 0; r < a.mr; ++r) for (int l = 0; l < a.ml; ++l) c.at(l, s, r, lab) += a.at(l, s, r, lab);
            const int lo = j == 1 ? 0 : a.ml, ro = j == x.N ? 0 : a.mr;
            for (int r = 0; r < b.mr; ++r) for (int l = 0; l < b.ml; ++l) c.at(lo + l, s, ro + r, lab) += b.at(l, s, r, lab);
        }
        z.A[j] = c;
    }
    return z;
}
// <x|y>
inline double overlap(const HostMPS& x, const HostMPS& y) {
    std::vector<double> E(1, 1.);                 // [ax][ay]
    int ax = 1, ay = 1;
    for (int j = 1; j <= x.N; ++j) {
        const Site &a = x.A[j], &b = y.A[j];
        if (a.L != b.L) throw std::runtime_error("overlap: Label index on different sites");
        std::vector<double> F((size_t)a.mr * b.mr, 0.);
        for (int lab = 0; lab < a.L; ++lab) for (int s = 0; s < 2; ++s)
            for (int rx = 0; rx < a.mr; ++rx) for (int ry = 0; ry < b.mr; ++ry) {
                double acc = 0.;
                for (int lx = 0; lx < ax; ++lx) { const double av = a.at(lx, s, rx, lab); if (av == 0.) continue;
                    for (int ly = 0; ly < ay; ++ly) acc += E[lx + (size_t)ax * ly] * av * b.at(ly, s, ry, lab); }
                F[rx + (size_t)a.mr * ry] += acc;
            }
        E.swap(F); ax = a.mr; ay = b.mr;
    }
    return E[0];
}
// orthogonalise from the right (QR by SVD without truncation), then truncate left to right with
// (cutoff, maxm): the "orthogonalize(args)" step behind ITensor's sum(psis,args)
inline void compress(HostMPS& psi, double cutoff, int maxm, bool center_first = false) {
    const int N = psi.N;
    auto split = [&](int j, bool to_left, bool trunc) {
        // to_left: A_j = (U S)(V^T): A_j <- V^T (right-orthonormal), A_{j-1} <- A_{j-1} U S
        // else   : A_j = U (S V^T): A_j <- U (left-orthonormal),   A_{j+1} <- S V^T A_{j+1}
        Site& a = psi.A[j];
        const int L = a.L;
        int R, C;
        std::vector<double> M;
        if (to_left) { R = a.ml; C = 2 * a.mr * L; M = a.a; }     // rows l, cols (s,r,lab): memory order already
        else {
            R = a.ml * 2 * L; C = a.mr; M.assign((size_t)R * C, 0.);
            for (int lab = 0; lab < L; ++lab) for (int r = 0; r < a.mr; ++r) for (int s = 0; s < 2; ++s) for (int l = 0; l < a.ml; ++l)
                M[(l + a.ml * (s + 2 * lab)) + (size_t)R * r] = a.at(l, s, r, lab);
        }
        std::vector<double> U, s, Vt;
        jacobi_svd(R, C, M, U, s, Vt);
        int k = (int)s.size();
        if (trunc) { std::vector<double> P(k); for (int g = 0; g < k; ++g) P[g] = s[g] * s[g]; k = truncate_rule(P, maxm, 1, cutoff); }
        const int k0 = (int)s.size();
        if (to_left) {
            Site na(k, a.mr, L);
            for (int col = 0; col < C; ++col) for (int g = 0; g < k; ++g) na.a[g + (size_t)k * col] = Vt[g + (size_t)k0 * col];
            Site& p = psi.A[j - 1];
            Site np(p.ml, k, p.L);
            for (int lab = 0; lab < p.L; ++lab) for (int g = 0; g < k; ++g) for (int sg = 0; sg < 2; ++sg) for (int l = 0; l < p.ml; ++l) {
                double acc = 0.; for (int r = 0; r < p.mr; ++r) acc += p.at(l, sg, r, lab) * U[r + (size_t)R * g] * s[g];
                np.at(l, sg, g, lab) = acc;
            }
            a = na; p = np;
        } else {
            Site na(a.ml, k, L);
            for (int lab = 0; lab < L; ++lab) for (int g = 0; g < k; ++g) for (int sg = 0; sg < 2; ++sg) for (int l = 0; l < a.ml; ++l)
                na.at(l, sg, g, lab) = U[(l + a.ml * (sg + 2 * lab)) + (size_t)R * g];
            Site& nx = psi.A[j + 1];
            Site nn(k, nx.mr, nx.L);
            for (int lab = 0; lab < nx.L; ++lab) for (int r = 0; r < nx.mr; ++r) for (int sg = 0; sg < 2; ++sg) for (int g = 0; g < k; ++g) {
                double acc = 0.; for (int l = 0; l < nx.ml; ++l) acc += s[g] * Vt[g + (size_t)k0 * l] * nx.at(l, sg, r, lab);
                nn.at(g, sg, r, lab) = acc;
            }
            a = na; nx = nn;
        }
    };
    for (int j = N; j >= 2; --j) split(j, true, false);
    for (int j = 1; j <= N - 1; ++j) split(j, false, true);
    if (center_first) for (int j = N; j >= 2; --j) split(j, true, false);      // W.position(1): orthogonality centre back to site 1
}
inline double norm_site(const Site& s) { double n2 = 0.; for (double v : s.a) n2 += v * v; return std::sqrt(n2); }

// ---- files ("TNMLW1" / "TNMLS1") -------------------------------------------------------------
inline void write_mps(const std::string& fname, const HostMPS& W) {
    std::ofstream f(fname, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("Couldn't open " + fname + " for writing");
    const char magic[8] = {'T', 'N', 'M', 'L', 'W', '1', 0, 0};
    f.write(magic, 8);
    const int32_t hdr[3] = {W.N, W.c0, 2};
    f.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
    for (int j = 1; j <= W.N; ++j) {
        const Site& s = W.A[j];
        const int32_t d[3] = {s.ml, s.mr, s.L};
        f.write(reinterpret_cast<const char*>(d), sizeof d);
        f.write(reinterpret_cast<const char*>(s.a.data()), sizeof(double) * s.a.size());
    }
    if (!f) throw std::runtime_error("write to " + fname + " failed");
}
inline HostMPS read_mps(const std::string& fname) {
    std::ifstream f(fname, std::ios::binary);
    if (!f) throw std::runtime_error("Couldn't open " + fname);
    char magic[8]; f.read(magic, 8);
    if (!f || std::memcmp(magic, "TNMLW1", 6) != 0) throw std::runtime_error(fname + " is not a TNMLW1 weight file");
    int32_t hdr[3]; f.read(reinterpret_cast<char*>(hdr), sizeof hdr);
    if (!f || hdr[0] < 1 || hdr[2] != 2) throw std::runtime_error(fname + ": bad header");
    HostMPS W(hdr[0]); W.c0 = hdr[1];
    for (int j = 1; j <= W.N; ++j) {
        int32_t d[3]; f.read(reinterpret_cast<char*>(d), sizeof d);
        if (!f || d[0] < 1 || d[1] < 1 || (d[2] != 1 && d[2] != NL)) throw std::runtime_error(fname + ": bad site record");
        W.A[j] = Site(d[0], d[1], d[2]);
        f.read(reinterpret_cast<char*>(W.A[j].a.data()), sizeof(double) * W.A[j].a.size());
        if (!f) throw std::runtime_error(fname + ": truncated");
    }
    return W;
}
inline void write_sites(const std::string& fname, int N, int d) {           // SiteSet(N,d), fixedL.cc:630-631
    std::ofstream f(fname, std::ios::binary | std::ios::trunc);
    const char magic[8] = {'T', 'N', 'M', 'L', 'S', '1', 0, 0};
    f.write(magic, 8);
    const int32_t hdr[2] = {N, d};
    f.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
}
inline void read_sites(const std::string& fname, int* N, int* d) {          // fixedL.cc:621-626
    std::ifstream f(fname, std::ios::binary);
    char magic[8]; int32_t hdr[2];
    f.read(magic, 8); f.read(reinterpret_cast<char*>(hdr), sizeof hdr);
    if (!f || std::memcmp(magic, "TNMLS1", 6) != 0) throw std::runtime_error(fname + " is not a TNMLS1 site file");
    *N = hdr[0]; *d = hdr[1];
}
inline bool file_exists(const std::string& f) { std::ifstream s(f); return (bool)s; }

}  // namespace tnmlh
