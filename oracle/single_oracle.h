/* single_oracle.h -- CPU oracle for the per-label variant (reference single.cc / single.h).
 *
 * TEST INFRASTRUCTURE ONLY, same rules as fixedl_oracle.h: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline may use it; nothing under tnml_amd/ does.  PARITY UNPINNED: the reference ships no tests or golden
 * vectors and cannot be built here (ITensor v2 absent); this restatement is cross-checked against an independent
 * numpy restatement (oracle/np_restatement.py) only.
 *
 * The model is a plain MPS W (no Label index); f(x_n) = W . Phi(x_n) is regressed onto y_n = [l_n == L]
 * (single.h:103,193).  fp64 throughout, dense per-image t.v (Precalc = true, the reference default). */
#ifndef SINGLE_ORACLE_H
#define SINGLE_ORACLE_H
#include "fixedl_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sorc sorc;

/* per-bond report of single.h:523-728 mldmrg */
typedef struct {
    int sweep, half, c;            /* "Sweep %d Half %d Bond %d" (single.h:566) */
    int origm, newm;
    double truncerr;
    double cost_old;               /* oC = quadcost(oB)  single.h:621 */
    double cost_cg;                /* C  = quadcost(B)   single.h:622 */
    double reg_cost;               /* lambda |B|^2       single.h:626 */
    double cost_after_svd;         /* newC               single.h:683 */
    double norm_oB, norm_newB;     /* Print(norm(..))    single.h:572,681 */
    int cg_skipped;                /* 1: "|r| < cconv, not optimizing" single.h:203-207 */
    orc_cg_trace cg;
} sorc_bond_report;

/* phi: [NT][N*2] as in orc_create; labels 0..9; target = the selected label L (single.cc:19) */
sorc* sorc_create(int N, int NT, const double* phi, const int* labels, int target, int nthread);
void sorc_destroy(sorc* o);
/* single.cc:71-84 feature maps from raw bytes with g = byte/255 (mllib/mnist.h:495), x = g/255:
   normal = [cos(pi x/2), sin(pi x/2)], series = [1, x/4] */
void sorc_features(int N, int NT, const unsigned char* pixels, int normal, double* phi);

int sorc_set_site(sorc* o, int j, int ml, int mr, const double* A);     /* A[ml][2][mr] column-major */
int sorc_site_dims(const sorc* o, int j, int* ml, int* mr);
int sorc_get_site(const sorc* o, int j, double* A);

int sorc_init(sorc* o);                                /* single.cc:181-199: E_N .. E_3 */
int sorc_set_bond(sorc* o, int b);                     /* single.h:581-596 (and single.cc:204-216 for b=1): dense t.v */
int sorc_shiftE(sorc* o, int b, int from_left);        /* single.h:688-710 */
int sorc_get_env(const sorc* o, int j, int i, double* E, int* m);

int sorc_bond_dims(const sorc* o, int b, int* mL, int* mR);
int sorc_bond_tensor(const sorc* o, int b, double* B); /* oB = W.A(c)*W.A(c+dc), single.h:570 */

int sorc_forward(const sorc* o, const double* B, double* P);            /* P[NT] = B*t.v */
int sorc_gradient(const sorc* o, const double* B, double* G);           /* sum_n dP_n * dag(t.v), single.h:185-198 */
double sorc_quadcost(const sorc* o, const double* B, double lambda, double* reg_cost);   /* single.h:82-112 */
/* single.h:162-288; returns 1 if it did not optimise (|r| < cconv at entry), 0 otherwise, <0 on error */
int sorc_cgrad(const sorc* o, double* B, int npass, double lambda, double cconv, orc_cg_trace* trace);
/* fast_cgrad (single.h:290-398): one pass over the images per CG step, residual by recurrence; same return values as sorc_cgrad */
int sorc_fast_cgrad(const sorc* o, double* B, int npass, double lambda, double cconv, orc_cg_trace* trace);
/* exact (single.h:117-160): B = y Phi^+ through the thin SVD of the D x NT matrix of the v_n, filtered inverse s/(s^2+lambda) above pcut */
int sorc_exact(const sorc* o, double* B, double lambda, double pcut);
/* optimiser used by sorc_mldmrg: 0 = conj (default), 1 = fast_conj, 2 = exact (single.h:598-600); pcut of the exact solver (default 1E-8) */
int sorc_set_method(sorc* o, int method);
int sorc_set_pcut(sorc* o, double pcut);
/* pinv (single.h:404-517) from the start V0 (D x r, column major; the reference's is random and time-seeded): B (out, D), ve[0..npass] the V*E
   trace, the passes run, the singular values of the last E */
int sorc_pinv(const sorc* o, const double* V0, int r, int npass, double lambda, double pcut, double* B, double* ve, int* npass_done, double* Dsv);
/* noise of the sweeps (single.cc:25,222): >= 1E-14 makes sorc_mldmrg split through the density matrix rho + noise * drho (single.h:648-672) */
int sorc_set_noise(sorc* o, double noise);
int sorc_noise_split(sorc* o, const double* B, int b, int ha, double noise, double cutoff, int maxm, int minm, double* truncerr, int* newm);
int sorc_svd_split(sorc* o, const double* B, int b, int ha, double cutoff, int maxm, int minm,
                   double* truncerr, int* newm, double* sv, int* nsv);   /* single.h:636-646 (noise = 0) */
int sorc_mldmrg(sorc* o, int nsweep, int maxm, int minm, double cutoff, int npass, double lambda,
                double cconv, int max_bonds, sorc_bond_report* reports);
/* decision function of image i: full contraction W . Phi(x_i)  (separate_fulltest.cc) */
int sorc_output(const sorc* o, int i, double* f);

#ifdef __cplusplus
}
#endif
#endif
