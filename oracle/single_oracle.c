/* single_oracle.c -- CPU restatement of the per-label variant of the reference:
 *   /root/reference/single.cc   main: features (71-84), initial projections (181-199), precalc (204-216)
 *   /root/reference/single.h    TState (19-25), quadcost (82-112), exact (117-160), cgrad (162-288), fast_cgrad (290-398), mldmrg (523-728)
 *                               incl. the density-matrix split with a noise term (648-672); pinv (404-517) from a given start
 *   /root/reference/paralleldo.h static chunking, fork-join
 * TEST INFRASTRUCTURE ONLY (see single_oracle.h).  PARITY UNPINNED (no reference tests, ITensor absent).
 *
 * Same conventions as fixedl_oracle.c: column-major tensors in ITensor index order, A_j[l][s][r],
 * B[a][s][t][be]; dense per-image t.v[a][s][t][be] (Precalc = true); per-thread partials summed in thread
 * order; scaleTo(1.) calls omitted (value preserving).  SVD and truncation rule are the
 * ones of fixedl_oracle.c (thin_svd, orc_truncate). */
#define _GNU_SOURCE
#include "single_oracle.h"

#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_thin_svd(int R, int C, const double* A, double* U, double* s, double* Vt);   /* fixedl_oracle.c */

typedef struct { int ml, mr; double* a; } ssite_t;
typedef struct { int m; double* e; } senv_t;            /* e = [NT][m] */

struct sorc {
    int N, NT, target, nthread;
    double* phi; int* labels;
    ssite_t* W;     /* 1..N */
    senv_t* E;      /* 1..N: TState::E */
    int currb;
    double* v; size_t vsz; int vmL, vmR;
    int sw, b, ha;
    int method;     /* 0 = conj (cgrad), 1 = fast_conj (fast_cgrad), 2 = exact: single.h:598-600 */
    double pcut;    /* PCut of the exact solver (single.cc: pcut key, default 1E-8) */
    double noise;   /* sweeps.noise() (single.cc:25,222): >= 1E-14 selects the density-matrix split of single.h:648-672 */
};

static int sfail(const char* msg) { fprintf(stderr, "single_oracle: %s\n", msg); return -1; }

typedef struct { size_t n, begin, end; } SBound;
typedef void (*stask_fn)(void* arg, SBound b);
typedef struct { stask_fn fn; void* arg; SBound b; } sthr_arg;
static void* sthr_main(void* p) { sthr_arg* t = (sthr_arg*)p; t->fn(t->arg, t->b); return NULL; }
/* single.cc:136-151 bounds + paralleldo.h:51-67 fork-join */
static void sparallel_do(int nthread, size_t ntask, stask_fn fn, void* arg) {
    SBound bounds[16]; pthread_t th[16]; sthr_arg ta[16];
    size_t th_size = ntask / (size_t)nthread, bcount = 0;
    for (int n = 0; n < nthread; ++n) { bounds[n].n = (size_t)n; bounds[n].begin = bcount; bounds[n].end = bcount + th_size; bcount += th_size; }
    bounds[nthread - 1].end = ntask;
    if (nthread == 1) { fn(arg, bounds[0]); return; }
    for (int n = 0; n < nthread; ++n) { ta[n].fn = fn; ta[n].arg = arg; ta[n].b = bounds[n]; pthread_create(&th[n], NULL, sthr_main, &ta[n]); }
    for (int n = 0; n < nthread; ++n) pthread_join(th[n], NULL);
}

void sorc_features(int N, int NT, const unsigned char* pixels, int normal, double* phi) {
    for (size_t k = 0; k < (size_t)NT * N; ++k) {
        const double g = pixels[k] / 255.;                   /* mllib/mnist.h:495 */
        const double x = g / 255.;                           /* single.cc:74 */
        if (normal) { phi[2 * k] = cos(M_PI / 2. * x); phi[2 * k + 1] = sin(M_PI / 2. * x); }   /* :77 */
        else        { phi[2 * k] = 1.; phi[2 * k + 1] = x / 4.; }                               /* :81 */
    }
}

sorc* sorc_create(int N, int NT, const double* phi, const int* labels, int target, int nthread) {
    if (N < 4 || NT < 1 || nthread < 1 || nthread > 16 || target < 0 || target > 9) { sfail("bad arguments"); return NULL; }
    sorc* o = (sorc*)calloc(1, sizeof *o);
    o->N = N; o->NT = NT; o->target = target; o->nthread = nthread;
    o->phi = (double*)malloc(sizeof(double) * (size_t)NT * N * 2);
    memcpy(o->phi, phi, sizeof(double) * (size_t)NT * N * 2);
    o->labels = (int*)malloc(sizeof(int) * (size_t)NT);
    memcpy(o->labels, labels, sizeof(int) * (size_t)NT);
    o->W = (ssite_t*)calloc((size_t)N + 2, sizeof(ssite_t));
    o->E = (senv_t*)calloc((size_t)N + 2, sizeof(senv_t));
    o->currb = -1; o->sw = 1; o->b = 1; o->ha = 1; o->pcut = 1E-8;
    return o;
}
void sorc_destroy(sorc* o) {
    if (!o) return;
    for (int j = 0; j <= o->N + 1; ++j) { free(o->W[j].a); free(o->E[j].e); }
    free(o->W); free(o->E); free(o->phi); free(o->labels); free(o->v); free(o);
}
int sorc_set_site(sorc* o, int j, int ml, int mr, const double* A) {
    if (j < 1 || j > o->N) return sfail("site out of range");
    ssite_t* s = &o->W[j];
    free(s->a);
    s->ml = ml; s->mr = mr;
    s->a = (double*)malloc(sizeof(double) * (size_t)ml * 2 * mr);
    memcpy(s->a, A, sizeof(double) * (size_t)ml * 2 * mr);
    o->currb = -1;
    return 0;
}
int sorc_site_dims(const sorc* o, int j, int* ml, int* mr) { if (j < 1 || j > o->N || !o->W[j].a) return sfail("site not set"); *ml = o->W[j].ml; *mr = o->W[j].mr; return 0; }
int sorc_get_site(const sorc* o, int j, double* A) { if (j < 1 || j > o->N || !o->W[j].a) return sfail("site not set"); memcpy(A, o->W[j].a, sizeof(double) * (size_t)o->W[j].ml * 2 * o->W[j].mr); return 0; }

static const double* sphi(const sorc* o, int i, int j) { return o->phi + ((size_t)i * o->N + (j - 1)) * 2; }
static double sy(const sorc* o, int i) { return o->labels[i] == o->target ? 1. : 0.; }     /* single.h:103,193 */

/* out[l] = sum_{s,r} phi[s] A[l,s,r] Ein[r]  (Ein NULL: chain end, mr == 1) */
static void step_from_right(const ssite_t* A, const double* ph, const double* Ein, double* out) {
    for (int l = 0; l < A->ml; ++l) {
        double acc = 0.;
        for (int r = 0; r < A->mr; ++r) {
            double t = ph[0] * A->a[l + (size_t)A->ml * (0 + 2 * r)] + ph[1] * A->a[l + (size_t)A->ml * (1 + 2 * r)];
            acc += t * (Ein ? Ein[r] : 1.);
        }
        out[l] = acc;
    }
}
static void step_from_left(const ssite_t* A, const double* ph, const double* Ein, double* out) {
    for (int r = 0; r < A->mr; ++r) {
        double acc = 0.;
        for (int l = 0; l < A->ml; ++l) {
            double t = ph[0] * A->a[l + (size_t)A->ml * (0 + 2 * r)] + ph[1] * A->a[l + (size_t)A->ml * (1 + 2 * r)];
            acc += (Ein ? Ein[l] : 1.) * t;
        }
        out[r] = acc;
    }
}
static void senv_alloc(sorc* o, int j, int m) { free(o->E[j].e); o->E[j].m = m; o->E[j].e = (double*)malloc(sizeof(double) * (size_t)o->NT * m); }

typedef struct { sorc* o; int site, prev, from_left; double* out; int mout; } senv_task;
static void senv_fn(void* p, SBound b) {
    senv_task* t = (senv_task*)p; sorc* o = t->o;
    const ssite_t* A = &o->W[t->site];
    for (size_t i = b.begin; i < b.end; ++i) {
        const double* Ein = t->prev ? o->E[t->prev].e + i * (size_t)o->E[t->prev].m : NULL;
        if (t->from_left) step_from_left(A, sphi(o, (int)i, t->site), Ein, t->out + i * (size_t)t->mout);
        else              step_from_right(A, sphi(o, (int)i, t->site), Ein, t->out + i * (size_t)t->mout);
    }
}
static int smake_env(sorc* o, int site, int prev, int from_left) {
    const ssite_t* A = &o->W[site];
    int mout = from_left ? A->mr : A->ml, min_ = from_left ? A->ml : A->mr;
    if (prev && o->E[prev].m != min_) return sfail("env dimension mismatch");
    if (!prev && min_ != 1) return sfail("chain end with outer dimension != 1");
    double* out = (double*)malloc(sizeof(double) * (size_t)o->NT * mout);
    senv_task t = { o, site, prev, from_left, out, mout };
    sparallel_do(o->nthread, (size_t)o->NT, senv_fn, &t);
    free(o->E[site].e); o->E[site].m = mout; o->E[site].e = out;
    return 0;
}
int sorc_init(sorc* o) {                                   /* single.cc:181-199 */
    for (int j = 1; j <= o->N; ++j) if (!o->W[j].a) return sfail("W not fully set");
    (void)senv_alloc;
    if (smake_env(o, o->N, 0, 0)) return -1;
    for (int j = o->N - 1; j >= 3; --j) if (smake_env(o, j, j + 1, 0)) return -1;
    o->currb = -1;
    return sorc_set_bond(o, 1);                            /* single.cc:204-216 */
}
int sorc_shiftE(sorc* o, int b, int from_left) {           /* single.h:688-710 with c = b (ha=1) or b+1 (ha=2) */
    int c = from_left ? b : b + 1, dc = from_left ? +1 : -1;
    if (c == 1 || c == o->N) return smake_env(o, c, 0, from_left);
    return smake_env(o, c, c - dc, from_left);
}
int sorc_get_env(const sorc* o, int j, int i, double* E, int* m) {
    if (j < 1 || j > o->N || !o->E[j].e) return sfail("env not built");
    if (m) *m = o->E[j].m;
    if (E) memcpy(E, o->E[j].e + (size_t)i * o->E[j].m, sizeof(double) * (size_t)o->E[j].m);
    return 0;
}

typedef struct { sorc* o; int b; } ssb_task;
static void ssb_fn(void* p, SBound bd) {                   /* single.h:581-596 */
    ssb_task* t = (ssb_task*)p; sorc* o = t->o; int b = t->b;
    int lc = b - 1, rc = b + 2;
    for (size_t i = bd.begin; i < bd.end; ++i) {
        const double* p1 = sphi(o, (int)i, b); const double* p2 = sphi(o, (int)i, b + 1);
        const double* LE = lc > 0 ? o->E[lc].e + i * (size_t)o->vmL : NULL;
        const double* RE = rc < o->N + 1 ? o->E[rc].e + i * (size_t)o->vmR : NULL;
        double* v = o->v + i * o->vsz;
        for (int be = 0; be < o->vmR; ++be) for (int tt = 0; tt < 2; ++tt) for (int s = 0; s < 2; ++s) for (int a = 0; a < o->vmL; ++a)
            v[a + (size_t)o->vmL * (s + 2 * (tt + 2 * be))] = (LE ? LE[a] : 1.) * p1[s] * p2[tt] * (RE ? RE[be] : 1.);
    }
}
int sorc_bond_dims(const sorc* o, int b, int* mL, int* mR) {
    if (b < 1 || b > o->N - 1 || !o->W[b].a || !o->W[b + 1].a) return sfail("bad bond");
    *mL = o->W[b].ml; *mR = o->W[b + 1].mr; return 0;
}
int sorc_set_bond(sorc* o, int b) {
    int mL, mR; if (sorc_bond_dims(o, b, &mL, &mR)) return -1;
    if (b - 1 > 0 && (!o->E[b - 1].e || o->E[b - 1].m != mL)) return sfail("left env missing or wrong size");
    if (b + 2 < o->N + 1 && (!o->E[b + 2].e || o->E[b + 2].m != mR)) return sfail("right env missing or wrong size");
    free(o->v);
    o->vmL = mL; o->vmR = mR; o->vsz = (size_t)mL * 4 * mR;
    o->v = (double*)malloc(sizeof(double) * o->vsz * (size_t)o->NT);
    ssb_task t = { o, b };
    sparallel_do(o->nthread, (size_t)o->NT, ssb_fn, &t);
    o->currb = b;
    return 0;
}
int sorc_bond_tensor(const sorc* o, int b, double* B) {    /* single.h:570 */
    int mL, mR; if (sorc_bond_dims(o, b, &mL, &mR)) return -1;
    const ssite_t* A1 = &o->W[b]; const ssite_t* A2 = &o->W[b + 1];
    if (A1->mr != A2->ml) return sfail("link mismatch");
    int k = A1->mr;
    for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t) for (int s = 0; s < 2; ++s) for (int a = 0; a < mL; ++a) {
        double acc = 0.;
        for (int g = 0; g < k; ++g) acc += A1->a[a + (size_t)mL * (s + 2 * g)] * A2->a[g + (size_t)k * (t + 2 * be)];
        B[a + (size_t)mL * (s + 2 * (t + 2 * be))] = acc;
    }
    return 0;
}
static double sdot(const double* x, const double* y, size_t n) { double s = 0.; for (size_t k = 0; k < n; ++k) s += x[k] * y[k]; return s; }
static double ssq(const double* x, size_t n) { return sdot(x, x, n); }

int sorc_forward(const sorc* o, const double* B, double* P) {
    if (!o->v) return sfail("setBond not called");
    for (int i = 0; i < o->NT; ++i) P[i] = sdot(B, o->v + (size_t)i * o->vsz, o->vsz);
    return 0;
}
/* sum_n dP_n*dag(t.v) with per-thread accumulators summed in thread order (single.h:185-199), optional sum dP^2 */
typedef struct { const sorc* o; const double* B; double* tensors; double* reals; } sgrad_task;
static void sgrad_fn(void* p, SBound b) {
    sgrad_task* t = (sgrad_task*)p; const sorc* o = t->o;
    double* T = t->tensors + b.n * o->vsz;
    for (size_t i = b.begin; i < b.end; ++i) {
        const double* v = o->v + i * o->vsz;
        double P = sdot(t->B, v, o->vsz);                  /* Bt.real() */
        double dP = sy(o, (int)i) - P;                     /* :193 */
        for (size_t k = 0; k < o->vsz; ++k) T[k] += dP * v[k];
        if (t->reals) t->reals[b.n] += dP * dP;            /* :258 */
    }
}
static void seval_gradient(const sorc* o, const double* B, double* out, double* csum) {
    double* tensors = (double*)calloc(o->vsz * (size_t)o->nthread, sizeof(double));
    double* reals = (double*)calloc((size_t)o->nthread, sizeof(double));
    sgrad_task t = { o, B, tensors, csum ? reals : NULL };
    sparallel_do(o->nthread, (size_t)o->NT, sgrad_fn, &t);
    memset(out, 0, sizeof(double) * o->vsz);
    for (int n = 0; n < o->nthread; ++n) for (size_t k = 0; k < o->vsz; ++k) out[k] += tensors[(size_t)n * o->vsz + k];   /* stdx::accumulate */
    if (csum) { double c = 0.; for (int n = 0; n < o->nthread; ++n) c += reals[n]; *csum = c; }
    free(tensors); free(reals);
}
int sorc_gradient(const sorc* o, const double* B, double* G) { if (!o->v) return sfail("setBond not called"); seval_gradient(o, B, G, NULL); return 0; }

typedef struct { const sorc* o; const double* B; double* reals; int pap; } sqc_task;
static void sqc_fn(void* p, SBound b) {
    sqc_task* t = (sqc_task*)p; const sorc* o = t->o;
    for (size_t i = b.begin; i < b.end; ++i) {
        double P = sdot(t->B, o->v + i * o->vsz, o->vsz);
        if (t->pap) t->reals[b.n] += P * P;                /* sqr(norm(pv)) :229 */
        else { double dP = sy(o, (int)i) - P; t->reals[b.n] += dP * dP; }   /* :103-104 */
    }
}
double sorc_quadcost(const sorc* o, const double* B, double lambda, double* reg_cost) {   /* single.h:82-112 */
    double* reals = (double*)calloc((size_t)o->nthread, sizeof(double));
    sqc_task t = { o, B, reals, 0 };
    sparallel_do(o->nthread, (size_t)o->NT, sqc_fn, &t);
    double C = 0.; for (int n = 0; n < o->nthread; ++n) C += reals[n];
    double CR = lambda * ssq(B, o->vsz);
    if (reg_cost) *reg_cost = CR;
    free(reals);
    return C + CR;
}
int sorc_cgrad(const sorc* o, double* B, int npass, double lambda, double cconv, orc_cg_trace* tr) {   /* single.h:162-288 */
    if (!o->v) return sfail("setBond not called");
    if (npass > 64) return sfail("npass > 64");
    size_t n = o->vsz;
    double* r = (double*)malloc(sizeof(double) * n); double* p = (double*)malloc(sizeof(double) * n); double* nr = (double*)malloc(sizeof(double) * n);
    double* reals = (double*)malloc(sizeof(double) * (size_t)o->nthread);
    if (tr) memset(tr, 0, sizeof *tr);
    int ret = 0;
    seval_gradient(o, B, r, NULL);                                         /* :184-199 */
    if (lambda != 0.) for (size_t k = 0; k < n; ++k) r[k] = r[k] - lambda * B[k];   /* :200 */
    if (sqrt(ssq(r, n)) < cconv) { ret = 1; goto done; }                   /* :202-206 "not optimizing" */
    memcpy(p, r, sizeof(double) * n);                                      /* :208 */
    for (int pass = 1; pass <= npass; ++pass) {                            /* :209 */
        for (int k = 0; k < o->nthread; ++k) reals[k] = 0.;
        sqc_task t = { o, p, reals, 1 };
        sparallel_do(o->nthread, (size_t)o->NT, sqc_fn, &t);               /* :219-233 */
        double pAp = 0.; for (int k = 0; k < o->nthread; ++k) pAp += reals[k];   /* :234 */
        pAp += lambda * ssq(p, n);                                         /* :235 */
        double a = ssq(r, n) / pAp;                                        /* :237 */
        for (size_t k = 0; k < n; ++k) B[k] = B[k] + a * p[k];             /* :238 */
        if (tr) { tr->npass_done = pass; tr->pAp[pass - 1] = pAp; tr->alpha[pass - 1] = a; }
        if (pass == npass) break;                                          /* :241 */
        double csum = 0.;
        seval_gradient(o, B, nr, &csum);                                   /* :243-262 */
        if (lambda != 0.) for (size_t k = 0; k < n; ++k) nr[k] = nr[k] - lambda * B[k];   /* :264 */
        double q = sqrt(ssq(nr, n)) / sqrt(ssq(r, n));
        double beta = q * q;                                               /* :265 */
        memcpy(r, nr, sizeof(double) * n);                                 /* :266 */
        double C = csum + lambda * ssq(B, n);                              /* :269-270 */
        double rn = sqrt(ssq(r, n));
        if (tr) { tr->cost[pass - 1] = C; tr->rnorm[pass - 1] = rn; }
        if (rn < cconv) { if (tr) tr->converged = 1; break; }              /* :273-277 */
        for (size_t k = 0; k < n; ++k) p[k] = r[k] + beta * p[k];          /* :284 */
    }
done:
    free(r); free(p); free(nr); free(reals);
    return ret;
}
/* fast_cgrad, single.h:290-398: ONE pass over the images per CG step -- p*t.v gives |p.v|^2 for pAp (:359) and, weighted back
 * onto t.v, the tensor A p (:360) -- and the residual follows the recurrence nr = r - a*Ap (:378) instead of being recomputed.
 * Reproduced as written, including ":379 nr = nr - lambda*B" (the full -lambda*B is subtracted again at every pass; the exact
 * recurrence would subtract a*lambda*p) and the absence of a cost print.  Trace: pAp, alpha, |r|; cost stays 0. */
typedef struct { const sorc* o; const double* p; double* tensors; double* reals; } sfast_task;
static void sfast_fn(void* q, SBound b) {
    sfast_task* t = (sfast_task*)q; const sorc* o = t->o;
    double* T = t->tensors + b.n * o->vsz;
    for (size_t i = b.begin; i < b.end; ++i) {
        const double* v = o->v + i * o->vsz;
        double pvr = sdot(t->p, v, o->vsz);                /* pv.real() :358 */
        t->reals[b.n] += pvr * pvr;                        /* :359 */
        for (size_t k = 0; k < o->vsz; ++k) T[k] += pvr * v[k];   /* :360 */
    }
}
int sorc_fast_cgrad(const sorc* o, double* B, int npass, double lambda, double cconv, orc_cg_trace* tr) {
    if (!o->v) return sfail("setBond not called");
    if (npass > 64) return sfail("npass > 64");
    size_t n = o->vsz;
    double* r = (double*)malloc(sizeof(double) * n); double* p = (double*)malloc(sizeof(double) * n); double* Ap = (double*)malloc(sizeof(double) * n);
    double* tensors = (double*)malloc(sizeof(double) * n * (size_t)o->nthread);
    double* reals = (double*)malloc(sizeof(double) * (size_t)o->nthread);
    if (tr) memset(tr, 0, sizeof *tr);
    int ret = 0;
    seval_gradient(o, B, r, NULL);                                         /* :311-325 */
    if (lambda != 0.) for (size_t k = 0; k < n; ++k) r[k] = r[k] - lambda * B[k];   /* :326 */
    if (sqrt(ssq(r, n)) < cconv) { ret = 1; goto done; }                   /* :328-332 "not optimizing" */
    memcpy(p, r, sizeof(double) * n);                                      /* :334 */
    for (int pass = 1; pass <= npass; ++pass) {                            /* :335 */
        memset(tensors, 0, sizeof(double) * n * (size_t)o->nthread);       /* :345-346 */
        for (int k = 0; k < o->nthread; ++k) reals[k] = 0.;
        sfast_task t = { o, p, tensors, reals };
        sparallel_do(o->nthread, (size_t)o->NT, sfast_fn, &t);             /* :347-363 */
        double pAp = 0.; for (int k = 0; k < o->nthread; ++k) pAp += reals[k];   /* :364 */
        pAp += lambda * ssq(p, n);                                         /* :365 */
        double a = ssq(r, n) / pAp;                                        /* :367 */
        for (size_t k = 0; k < n; ++k) B[k] = B[k] + a * p[k];             /* :368 */
        if (tr) { tr->npass_done = pass; tr->pAp[pass - 1] = pAp; tr->alpha[pass - 1] = a; }
        if (pass == npass) break;                                          /* :371-375 */
        memset(Ap, 0, sizeof(double) * n);
        for (int th = 0; th < o->nthread; ++th) for (size_t k = 0; k < n; ++k) Ap[k] += tensors[(size_t)th * n + k];   /* :377 stdx::accumulate */
        double rn_old = sqrt(ssq(r, n));
        for (size_t k = 0; k < n; ++k) r[k] = r[k] - a * Ap[k];            /* :378 nr = r - a*Ap */
        if (lambda != 0.) for (size_t k = 0; k < n; ++k) r[k] = r[k] - lambda * B[k];   /* :379 (as written) */
        double rn = sqrt(ssq(r, n));
        double q = rn / rn_old;
        double beta = q * q;                                               /* :381 */
        if (tr) tr->rnorm[pass - 1] = rn;                                  /* :382 r = nr */
        if (rn < cconv) { if (tr) tr->converged = 1; break; }              /* :385-389 */
        for (size_t k = 0; k < n; ++k) p[k] = r[k] + beta * p[k];          /* :395 */
    }
done:
    free(r); free(p); free(Ap); free(tensors); free(reals);
    return ret;
}
/* exact, single.h:117-160: B = y Phi^+ with Phi = [v_1 ... v_NT] (D x NT, D = 4 mL mR), through the thin SVD of Phi and the
 * filtered inverse s -> s/(s^2 + lambda) for s > pcut, 0 otherwise (:145-153).  "Only works for rather small number of training
 * samples" (:114).  The incoming B is ignored.  [ITensor-recall] svd(Phi,U,S,V) with default arguments does not truncate. */
int sorc_exact(const sorc* o, double* B, double lambda, double pcut) {
    if (!o->v) return sfail("setBond not called");
    const int D = (int)o->vsz, NT = o->NT;
    const int k = D < NT ? D : NT;
    double* Phi = (double*)malloc(sizeof(double) * (size_t)D * NT);           /* column n = v_n (:137) */
    for (int n = 0; n < NT; ++n) memcpy(Phi + (size_t)D * n, o->v + (size_t)n * o->vsz, sizeof(double) * (size_t)D);
    double* U = (double*)malloc(sizeof(double) * (size_t)D * k); double* sv = (double*)malloc(sizeof(double) * (size_t)k);
    double* Vt = (double*)malloc(sizeof(double) * (size_t)k * NT);
    orc_thin_svd(D, NT, Phi, U, sv, Vt);                                     /* Phi = U diag(s) Vt; ITensor's U carries the image index: roles swapped, same product */
    memset(B, 0, sizeof(double) * (size_t)D);
    for (int g = 0; g < k; ++g) {
        const double s1 = sv[g];
        const double f = s1 > pcut ? s1 / (s1 * s1 + lambda) : 0.;           /* pseudoInv :145-153 */
        if (f == 0.) continue;
        double yv = 0.;
        for (int n = 0; n < NT; ++n) yv += sy(o, n) * Vt[g + (size_t)k * n];  /* yL * U */
        for (int i = 0; i < D; ++i) B[i] += yv * f * U[i + (size_t)D * g];    /* ... * Sinv * V (:158-159) */
    }
    free(Phi); free(U); free(sv); free(Vt);
    return 0;
}
/* pinv (single.h:404-517): the "truncated pseudo inverse solver" -- a subspace iteration on A = sum_n v_n v_n^T from an r-dimensional
 * start V (r = Ntarget): E = V^T A (:469-473,482-486), V <- polar factor of E through its SVD E = F D G (:492-495), until V*E = sum(D)
 * moves by less than 1E-4 (:500) or Npass passes; then B = yUS * Einv with Einv = F pseudoInv(D) G (:510), yUS = sum over the images of
 * the target label of v_n * V (:513-518).  In the reference the start is random(...) from a time-seeded generator (:457) and the result is
 * only PRINTED (its cost, single.h:596-601: the update that follows is cgrad on the untouched B), so there is nothing to be bit-compatible
 * with: here the start V0 (D x r, column major) is an argument.  ve[0] = the initial V*E, ve[p] = V*E after pass p; Dsv = the last D. */
int sorc_pinv(const sorc* o, const double* V0, int r, int npass, double lambda, double pcut, double* B, double* ve, int* npass_done, double* Dsv) {
    if (!o->v) return sfail("setBond not called");
    const int D = (int)o->vsz, NT = o->NT;
    if (r < 1 || r > D) return sfail("pinv: Ntarget out of range");
    double* V = (double*)malloc(sizeof(double) * (size_t)D * r); double* E = (double*)malloc(sizeof(double) * (size_t)D * r);   /* column k = V_k, E_k */
    double* U = (double*)malloc(sizeof(double) * (size_t)D * r); double* sv = (double*)malloc(sizeof(double) * (size_t)r);
    double* Wt = (double*)malloc(sizeof(double) * (size_t)r * r); double* pk = (double*)malloc(sizeof(double) * (size_t)NT);
    /* V = polarU(V0) (:458): V0 = U S Wt -> U Wt */
    orc_thin_svd(D, r, V0, U, sv, Wt);
    for (int k = 0; k < r; ++k) for (int d = 0; d < D; ++d) { double t = 0.; for (int g = 0; g < r; ++g) t += U[d + (size_t)D * g] * Wt[g + (size_t)r * k]; V[d + (size_t)D * k] = t; }
#define SORC_MAKE_E() do { \
        for (int k = 0; k < r; ++k) { \
            for (int n = 0; n < NT; ++n) pk[n] = sdot(V + (size_t)D * k, o->v + (size_t)n * o->vsz, (size_t)D); \
            double* Ek = E + (size_t)D * k; memset(Ek, 0, sizeof(double) * (size_t)D); \
            for (int n = 0; n < NT; ++n) { const double* vn = o->v + (size_t)n * o->vsz; for (int d = 0; d < D; ++d) Ek[d] += pk[n] * vn[d]; } \
        } } while (0)
    SORC_MAKE_E();
    double last = sdot(V, E, (size_t)D * r);                          /* :475 lastVE */
    if (ve) ve[0] = last;
    int done = 0;
    for (int pass = 1; pass <= npass; ++pass) {
        SORC_MAKE_E();                                                /* :482-486 */
        orc_thin_svd(D, r, E, U, sv, Wt);                             /* E^T (D x r) = U S Wt: F[a][g] = Wt[g][a], G[g][:] = U[:][g]  (:492) */
        for (int a = 0; a < r; ++a) for (int d = 0; d < D; ++d) { double t = 0.; for (int g = 0; g < r; ++g) t += Wt[g + (size_t)r * a] * U[d + (size_t)D * g]; V[d + (size_t)D * a] = t; }   /* :495 */
        const double VE = sdot(V, E, (size_t)D * r);                  /* :497 */
        done = pass;
        if (ve) ve[pass] = VE;
        if (fabs(VE - last) < 1E-4) break;                            /* :500 */
        last = VE;
    }
    if (npass_done) *npass_done = done;
    memset(B, 0, sizeof(double) * (size_t)D);
    if (done > 0) {
        if (Dsv) memcpy(Dsv, sv, sizeof(double) * (size_t)r);
        /* yUS (:513-518) with the V of the last pass; B = yUS * F pseudoInv(D) G (:510,519) */
        for (int a = 0; a < r; ++a) {
            double yus = 0.;
            for (int n = 0; n < NT; ++n) if (o->labels[n] == o->target) yus += sdot(V + (size_t)D * a, o->v + (size_t)n * o->vsz, (size_t)D);
            for (int g = 0; g < r; ++g) {
                const double f = sv[g] > pcut ? sv[g] / (sv[g] * sv[g] + lambda) : 0.;      /* :417-421 */
                const double cf = yus * Wt[g + (size_t)r * a] * f;
                if (cf != 0.) for (int d = 0; d < D; ++d) B[d] += cf * U[d + (size_t)D * g];
            }
        }
    }
#undef SORC_MAKE_E
    free(V); free(E); free(U); free(sv); free(Wt); free(pk);
    return 0;
}
int sorc_set_pcut(sorc* o, double pcut) { o->pcut = pcut; return 0; }
int sorc_set_noise(sorc* o, double noise) { if (!(noise >= 0.)) return sfail("noise must be >= 0"); o->noise = noise; return 0; }
int sorc_set_method(sorc* o, int method) { if (method < 0 || method > 2) return sfail("method must be 0 (conj), 1 (fast_conj) or 2 (exact)"); o->method = method; return 0; }
/* svd(B,U,S,V,svd_args) with U on the indices of W.A(c); W.A(c) = U, W.A(c+dc) = S*V  (single.h:636-646) */
int sorc_svd_split(sorc* o, const double* B, int b, int ha, double cutoff, int maxm, int minm,
                   double* truncerr, int* newm, double* sv_out, int* nsv) {
    int mL, mR; if (sorc_bond_dims(o, b, &mL, &mR)) return -1;
    int nl = 2 * mL, nr = 2 * mR;
    int R = ha == 1 ? nl : nr, C = ha == 1 ? nr : nl;
    double* M = (double*)malloc(sizeof(double) * (size_t)R * C);
    for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t) for (int s = 0; s < 2; ++s) for (int a = 0; a < mL; ++a) {
        int il = a + mL * s, ir = t + 2 * be;
        double x = B[a + (size_t)mL * (s + 2 * (t + 2 * be))];
        if (ha == 1) M[il + (size_t)R * ir] = x; else M[ir + (size_t)R * il] = x;
    }
    int k = R < C ? R : C;
    double* U = (double*)malloc(sizeof(double) * (size_t)R * k); double* s = (double*)malloc(sizeof(double) * (size_t)k);
    double* Vt = (double*)malloc(sizeof(double) * (size_t)k * C); double* P = (double*)malloc(sizeof(double) * (size_t)k);
    orc_thin_svd(R, C, M, U, s, Vt);
    for (int g = 0; g < k; ++g) P[g] = s[g] * s[g];
    double te = 0.;
    int m = orc_truncate(P, k, maxm, minm, cutoff, &te);
    if (truncerr) *truncerr = te;
    if (newm) *newm = m;
    if (nsv) *nsv = k;
    if (sv_out) memcpy(sv_out, s, sizeof(double) * (size_t)k);
    ssite_t* Sl = &o->W[b]; ssite_t* Sr = &o->W[b + 1];
    free(Sl->a); free(Sr->a);
    Sl->ml = mL; Sl->mr = m; Sr->ml = m; Sr->mr = mR;
    Sl->a = (double*)malloc(sizeof(double) * (size_t)mL * 2 * m);
    Sr->a = (double*)malloc(sizeof(double) * (size_t)m * 2 * mR);
    for (int g = 0; g < m; ++g) {
        for (int sI = 0; sI < 2; ++sI) for (int a = 0; a < mL; ++a) {
            int il = a + mL * sI;
            Sl->a[a + (size_t)mL * (sI + 2 * g)] = ha == 1 ? U[il + (size_t)R * g] : s[g] * Vt[g + (size_t)k * il];
        }
        for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t) {
            int ir = t + 2 * be;
            Sr->a[g + (size_t)m * (t + 2 * be)] = ha == 1 ? s[g] * Vt[g + (size_t)k * ir] : U[ir + (size_t)R * g];
        }
    }
    free(M); free(U); free(s); free(Vt); free(P);
    o->currb = -1;
    return 0;
}
/* the split of single.h:648-672 (noise >= 1E-14): rho = B B^dag over the indices of site c (c = b for ha = 1, b + 1 for ha = 2),
 *   drho = sum_n dr_n dr_n^dag with dr_n = (B * E_n) (x) E_n, E_n the environment on the outer link of site c (single.h:655-664; at the
 *   chain ends there is no such environment and dr_n = B for every image), rho += noise * drho, diagHermitian(rho) with the truncation
 *   parameters of the sweep (single.h:666), W_c = UU, W_{c+dc} = UU * B (single.h:667-668).  The truncation rule acts on the eigenvalues
 *   of rho as it acts on the squared singular values in sorc_svd_split [ITensor-recall, as there]. */
int sorc_noise_split(sorc* o, const double* B, int b, int ha, double noise, double cutoff, int maxm, int minm, double* truncerr, int* newm) {
    int mL, mR; if (sorc_bond_dims(o, b, &mL, &mR)) return -1;
    const int nl = 2 * mL, nr = 2 * mR;
    const int R = ha == 1 ? nl : nr, C = ha == 1 ? nr : nl;          /* rows: the indices of site c */
    double* M = (double*)malloc(sizeof(double) * (size_t)R * C);
    for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t) for (int s = 0; s < 2; ++s) for (int a = 0; a < mL; ++a) {
        const int il = a + mL * s, ir = t + 2 * be;
        const double x = B[a + (size_t)mL * (s + 2 * (t + 2 * be))];
        if (ha == 1) M[il + (size_t)R * ir] = x; else M[ir + (size_t)R * il] = x;
    }
    double* rho = (double*)calloc((size_t)R * R, sizeof(double));
    double* drho = (double*)calloc((size_t)R * R, sizeof(double));
    for (int j = 0; j < R; ++j) for (int i = 0; i < R; ++i) { double t = 0.; for (int y = 0; y < C; ++y) t += M[i + (size_t)R * y] * M[j + (size_t)R * y]; rho[i + (size_t)R * j] = t; }   /* :651-653 */
    const int c = ha == 1 ? b : b + 1;
    const int envsite = ha == 1 ? c - 1 : c + 1;
    const int have_env = ha == 1 ? c > 1 : c < o->N - 1;              /* :650,655 as written: "ha == 2 && c < N-1" */
    if (!have_env) {
        for (size_t k = 0; k < (size_t)R * R; ++k) drho[k] = (double)o->NT * rho[k];      /* dr = B for every image */
    } else {
        const int mE = ha == 1 ? mL : mR;
        if (o->E[envsite].m != mE || !o->E[envsite].e) { free(M); free(rho); free(drho); return sfail("noise split: environment missing"); }
        double* T = (double*)malloc(sizeof(double) * (size_t)2 * C);  /* (B * E_n)[site index of c][other indices] */
        for (int n = 0; n < o->NT; ++n) {
            const double* E = o->E[envsite].e + (size_t)n * mE;
            /* row index of M: il = a + mL s (ha = 1), ir = t + 2 be (ha = 2): contract the link of site c with E_n */
            for (int sc = 0; sc < 2; ++sc) for (int y = 0; y < C; ++y) {
                double t = 0.;
                for (int e = 0; e < mE; ++e) { const int row = ha == 1 ? e + mL * sc : sc + 2 * e; t += E[e] * M[row + (size_t)R * y]; }
                T[sc + 2 * (size_t)y] = t;
            }
            double w[2][2] = {{0., 0.}, {0., 0.}};
            for (int y = 0; y < C; ++y) for (int s1 = 0; s1 < 2; ++s1) for (int s2 = 0; s2 < 2; ++s2) w[s1][s2] += T[s1 + 2 * (size_t)y] * T[s2 + 2 * (size_t)y];
            for (int s2 = 0; s2 < 2; ++s2) for (int e2 = 0; e2 < mE; ++e2) for (int s1 = 0; s1 < 2; ++s1) for (int e1 = 0; e1 < mE; ++e1) {
                const int r1 = ha == 1 ? e1 + mL * s1 : s1 + 2 * e1, r2 = ha == 1 ? e2 + mL * s2 : s2 + 2 * e2;
                drho[r1 + (size_t)R * r2] += w[s1][s2] * E[e1] * E[e2];                   /* :664-665 */
            }
        }
        free(T);
    }
    for (size_t k = 0; k < (size_t)R * R; ++k) rho[k] += noise * drho[k];                 /* :665 */
    double* U = (double*)malloc(sizeof(double) * (size_t)R * R); double* ev = (double*)malloc(sizeof(double) * (size_t)R);
    double* Vt = (double*)malloc(sizeof(double) * (size_t)R * R);
    orc_thin_svd(R, R, rho, U, ev, Vt);                               /* symmetric positive semidefinite: singular values = eigenvalues, U = eigenvectors */
    double te = 0.;
    const int m = orc_truncate(ev, R, maxm, minm, cutoff, &te);       /* :666 diagHermitian(rho,UU,DD,svd_args) */
    if (truncerr) *truncerr = te;
    if (newm) *newm = m;
    ssite_t* Sl = &o->W[b]; ssite_t* Sr = &o->W[b + 1];
    free(Sl->a); free(Sr->a);
    Sl->ml = mL; Sl->mr = m; Sr->ml = m; Sr->mr = mR;
    Sl->a = (double*)malloc(sizeof(double) * (size_t)mL * 2 * m);
    Sr->a = (double*)malloc(sizeof(double) * (size_t)m * 2 * mR);
    for (int g = 0; g < m; ++g) {
        /* UU on site c (:668), UU * B on the other site (:667): (U^T M)[g][y] */
        for (int i = 0; i < R; ++i) {
            if (ha == 1) Sl->a[(i % mL) + (size_t)mL * ((i / mL) + 2 * g)] = U[i + (size_t)R * g];
            else         Sr->a[g + (size_t)m * i] = U[i + (size_t)R * g];
        }
        for (int y = 0; y < C; ++y) {
            double t = 0.; for (int i = 0; i < R; ++i) t += U[i + (size_t)R * g] * M[i + (size_t)R * y];
            if (ha == 1) Sr->a[g + (size_t)m * y] = t;
            else         Sl->a[(y % mL) + (size_t)mL * ((y / mL) + 2 * g)] = t;
        }
    }
    free(M); free(rho); free(drho); free(U); free(ev); free(Vt);
    o->currb = -1;
    return 0;
}
int sorc_mldmrg(sorc* o, int nsweep, int maxm, int minm, double cutoff, int npass, double lambda,
                double cconv, int max_bonds, sorc_bond_report* reports) {     /* single.h:523-728, Method = conj | fast_conj | exact (sorc_set_method), noise (sorc_set_noise) */
    int done = 0;
    while (o->sw <= nsweep) {
        while (o->ha <= 2) {
            if (max_bonds > 0 && done >= max_bonds) return done;
            int b = o->b, ha = o->ha;
            sorc_bond_report* rp = &reports[done];
            memset(rp, 0, sizeof *rp);
            rp->sweep = o->sw; rp->half = ha; rp->c = ha == 1 ? b : b + 1;      /* :558,566 */
            int mL, mR; if (sorc_bond_dims(o, b, &mL, &mR)) return -1;
            size_t n = (size_t)mL * 4 * mR;
            rp->origm = o->W[b].mr;                                            /* :569 */
            double* oB = (double*)malloc(sizeof(double) * n); double* B = (double*)malloc(sizeof(double) * n);
            if (sorc_bond_tensor(o, b, oB)) return -1;                         /* :570 */
            rp->norm_oB = sqrt(ssq(oB, n));                                    /* :572 */
            memcpy(B, oB, sizeof(double) * n);
            if (sorc_set_bond(o, b)) return -1;                                /* :579-596 */
            int rc = o->method == 2 ? sorc_exact(o, B, lambda, o->pcut)                      /* :600 */
                   : o->method == 1 ? sorc_fast_cgrad(o, B, npass, lambda, cconv, &rp->cg)   /* :599 */
                                    : sorc_cgrad(o, B, npass, lambda, cconv, &rp->cg);       /* :598 */
            if (rc < 0) return -1;
            rp->cg_skipped = rc;
            rp->cost_old = sorc_quadcost(o, oB, lambda, NULL);                 /* :621 */
            rp->cost_cg = sorc_quadcost(o, B, lambda, &rp->reg_cost);          /* :622,626 */
            if (o->noise < 1E-14) { if (sorc_svd_split(o, B, b, ha, cutoff, maxm, minm, &rp->truncerr, &rp->newm, NULL, NULL)) return -1; }   /* :626-646 */
            else if (sorc_noise_split(o, B, b, ha, o->noise, cutoff, maxm, minm, &rp->truncerr, &rp->newm)) return -1;                      /* :647-672 */
            double* newB = (double*)malloc(sizeof(double) * n);
            if (sorc_bond_tensor(o, b, newB)) return -1;                       /* :680 */
            rp->norm_newB = sqrt(ssq(newB, n));
            rp->cost_after_svd = sorc_quadcost(o, newB, lambda, NULL);         /* :683 */
            if (sorc_shiftE(o, b, ha == 1)) return -1;                         /* :688-710 */
            free(oB); free(B); free(newB);
            ++done;
            orc_sweepnext(&o->b, &o->ha, o->N);
        }
        o->sw += 1; o->b = 1; o->ha = 1;
    }
    return done;
}
int sorc_output(const sorc* o, int i, double* f) {
    int cap = 1; for (int j = 1; j <= o->N; ++j) { if (o->W[j].ml > cap) cap = o->W[j].ml; if (o->W[j].mr > cap) cap = o->W[j].mr; }
    double* cur = (double*)malloc(sizeof(double) * (size_t)cap); double* nxt = (double*)malloc(sizeof(double) * (size_t)cap);
    step_from_right(&o->W[o->N], sphi(o, i, o->N), NULL, cur);
    for (int j = o->N - 1; j >= 1; --j) { step_from_right(&o->W[j], sphi(o, i, j), cur, nxt); double* t = cur; cur = nxt; nxt = t; }
    *f = cur[0];
    free(cur); free(nxt);
    return 0;
}
