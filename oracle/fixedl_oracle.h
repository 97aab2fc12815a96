/*
 * fixedl_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a dependency-free fp64 restatement of the reference's fixedL two-site sweep
 * (reference: fixedL.cc, paralleldo.h, util.h at /root/reference).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and its arithmetic back-end
 * (ITensor v2) is not available offline, so nothing external pins this restatement; it is
 * cross-checked against an independent numpy restatement (oracle/np_restatement.py).
 *
 * Tensor layouts (column-major, first index fastest, 0-based offsets, ITensor index order):
 *   site tensor  A_j [ml][2][mr]      (+[10] last, only on the label site c0 = N/2)
 *   bond tensor  B   [mL][2][2][mR]   (+[10] last, only when c0 is b or b+1)
 *   environment  E_j [m]  or [m][10]  per image
 * Sites and bonds are 1-indexed exactly as in the reference.
 */
#ifndef FIXEDL_ORACLE_H
#define FIXEDL_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NL 10

typedef struct orc orc;

/* per-CG-pass trace, mirrors the prints of fixedL.cc:429-439 */
typedef struct {
    int npass_done;      /* number of passes entered */
    int converged;       /* 1 if |r| < cconv triggered (fixedL.cc:432) */
    double cost[64];     /* C (un-normalised) printed at :429, index pass-1 (not set for the last pass, :409) */
    double rnorm[64];    /* |r| printed at :434/:439 */
    double pAp[64];
    double alpha[64];
} orc_cg_trace;

/* per-bond report of mldmrg, mirrors the prints of fixedL.cc:523-533 */
typedef struct {
    int sweep, half, bond, c;      /* "Sweep %d Half %d Bond %d" prints c, not b (fixedL.cc:490) */
    int origm, newm;
    double truncerr;
    double norm_newB, diff_B_newB;
    double cost_after_svd;         /* un-normalised C returned by quadcost at :532 */
    double label_cost[ORC_NL];
    double reg_cost;
    long ncorrect;
    orc_cg_trace cg;
} orc_bond_report;

/* phi: [NT][N*2] feature values data[(j-1)*2 + (n-1)] = phi(pixel_j, n)  (fixedL.cc:28-47) */
orc* orc_create(int N, int NT, const double* phi, const int* labels, int nthread, int nbatch);
void orc_destroy(orc* o);
const char* orc_last_error(void);

/* reference feature map phi(g,n) = ((g/255)/4)^(n-1), g = pixel/255 (fixedL.cc:637-642 after
   mllib/mnist.h:495): fills phi[NT][N*2] from raw bytes */
void orc_features_series(int N, int NT, const unsigned char* pixels, double* phi);

/* weight MPS */
int orc_set_site(orc* o, int j, int ml, int mr, int has_label, const double* A);
int orc_site_dims(const orc* o, int j, int* ml, int* mr, int* has_label);
int orc_get_site(const orc* o, int j, double* A);

/* TrainStates::init / setBond / shiftE  (fixedL.cc:122-233) */
int orc_init(orc* o);
int orc_set_bond(orc* o, int b);
int orc_shiftE(orc* o, int b, int from_left);
/* environment of image i at site j: dims and copy-out ([m] or [m][10]) */
int orc_env_dims(const orc* o, int j, int* m, int* has_label);
int orc_get_env(const orc* o, int j, int i, double* E);

/* bond tensor oB = W.A(b)*W.A(b+1)  (fixedL.cc:494) */
int orc_bond_dims(const orc* o, int b, int* mL, int* mR, int* label_on_B);
int orc_bond_tensor(const orc* o, int b, double* B);

/* per-image model output P_n = B*t.v (fixedL.cc:318) for the current bond: P[NT][10] */
int orc_forward(const orc* o, const double* B, double* P);
/* gradient accumulator sum_n dP_n*dag(t.v) (fixedL.cc:375-385), same layout as B */
int orc_gradient(const orc* o, const double* B, double* G);

/* quadcost (fixedL.cc:280-344): returns C = sum_l C_l + lambda |B|^2 (un-normalised) */
double orc_quadcost(const orc* o, const double* B, double lambda,
                    double label_cost[ORC_NL], double* reg_cost, long* ncorrect);

/* cgrad (fixedL.cc:349-445): B updated in place */
int orc_cgrad(const orc* o, double* B, int npass, double lambda, double cconv, orc_cg_trace* trace);

/* ITensor truncate() rule as recalled in SURVEY.md 8(a9): p sorted descending, returns kept m */
int orc_truncate(const double* p, int n, int maxm, int minm, double cutoff, double* truncerr);

/* svd + "*= S" of fixedL.cc:519-521; ha = 1 (sweeping right, c=b) or 2 (sweeping left, c=b+1).
   sv (nullable) receives all singular values (descending), nsv their count. */
int orc_svd_split(orc* o, const double* B, int b, int ha, double cutoff, int maxm, int minm,
                  double* truncerr, int* newm, double* sv, int* nsv);

/* sweepnext of ITensor as recalled in SURVEY.md 8(a12) */
void orc_sweepnext(int* b, int* ha, int N);

/* mldmrg (fixedL.cc:451-570): runs at most max_bonds bond updates (<=0: all of nsweep sweeps),
   continuing from the bond position stored in the oracle; fills reports[0..ret). */
int orc_mldmrg(orc* o, int nsweep, int maxm, int minm, double cutoff, int npass, double lambda,
               double cconv, int max_bonds, orc_bond_report* reports, int verbose);

/* full contraction of image i with W (util.h:19-40 toverlap, centre = label site): out[10] */
int orc_toverlap(const orc* o, int i, double* out);

#ifdef __cplusplus
}
#endif
#endif
