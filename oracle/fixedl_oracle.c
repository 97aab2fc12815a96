/*
 * fixedl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into the product).
 *
 * A dependency-free fp64 restatement of the reference's fixedL two-site DMRG-style sweep:
 *   /root/reference/fixedL.cc      TState, TrainStates{init,setBond,shiftE,execute},
 *                                  quadcost, cgrad, mldmrg
 *   /root/reference/paralleldo.h   Bound / ParallelDo chunking and fork-join
 *   /root/reference/util.h         argmax (first maximum), toverlap
 * Each function cites the lines it follows.  The algorithm is kept as the reference has it:
 * the dense per-image effective tensor t.v is materialised by set_bond and every pass is a
 * dense contraction against it (this is what makes the reference DRAM-bound).
 *
 * What is NOT restated: ITensor's lazy scale bookkeeping (scaleTo(1.) is value preserving,
 * SURVEY.md 9-Q5), the proj_images/ disk spill (envs stay in RAM; I/O, not arithmetic), and
 * ITensor's SVD implementation (eig of M M^T with refinement) -- a one-sided Jacobi SVD gives
 * the same singular values / subspaces; the truncation rule follows SURVEY.md 8(a9).
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and ITensor is unavailable
 * offline, so this restatement is cross-checked only against the independent numpy
 * restatement in oracle/np_restatement.py (tests/test_oracle.py).
 */
#include "fixedl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NL ORC_NL

typedef struct { int ml, mr, L; double* a; } site_t;   /* [ml][2][mr][L] */
typedef struct { int m, L; double* e; } envs_t;         /* per image [m][L]; e = [NT][m*L] */

struct orc {
    int N, NT, c0;
    int nthread, nbatch, batchsize;
    double* phi;      /* [NT][N*2]  TState::data, fixedL.cc:39-46 */
    int* labels;      /* TState::l */
    site_t* W;        /* 1..N */
    envs_t* E;        /* 1..N, one slot per site exactly like the files B%03dE%05d (fixedL.cc:257-261) */
    int currb;        /* TrainStates::currb_ */
    double* v;        /* dense t.v for all images, image stride vsz */
    size_t vsz;       /* mL*4*mR*vL */
    int vmL, vmR, vL;
    /* sweep position for orc_mldmrg */
    int sw, b, ha;
};

static char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int fail(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

/* ------------------------------------------------------------------ ParallelDo ---------- */
/* paralleldo.h:8-19 */
typedef struct { size_t n, begin, end; } Bound;
typedef void (*task_fn)(void* arg, Bound b);
typedef struct { task_fn fn; void* arg; Bound b; } thr_arg;
static void* thr_main(void* p) { thr_arg* t = (thr_arg*)p; t->fn(t->arg, t->b); return NULL; }

/* paralleldo.h:32-43 (bounds) and :51-67 (fork-join); the reference caps at 16 futures (:55-56) */
static void parallel_do(int nthread, size_t ntask, task_fn fn, void* arg) {
    Bound bounds[16];
    pthread_t th[16];
    thr_arg ta[16];
    size_t th_size = ntask / (size_t)nthread, bcount = 0;
    for (int n = 0; n < nthread; ++n) {
        bounds[n].n = (size_t)n; bounds[n].begin = bcount; bounds[n].end = bcount + th_size;
        bcount += th_size;
    }
    bounds[nthread - 1].end = ntask;
    if (nthread == 1) { fn(arg, bounds[0]); return; }
    for (int n = 0; n < nthread; ++n) {
        ta[n].fn = fn; ta[n].arg = arg; ta[n].b = bounds[n];
        pthread_create(&th[n], NULL, thr_main, &ta[n]);
    }
    for (int n = 0; n < nthread; ++n) pthread_join(th[n], NULL);
}

/* ------------------------------------------------------------------ create / destroy ---- */
void orc_features_series(int N, int NT, const unsigned char* pixels, double* phi) {
    /* mllib/mnist.h:495 stores img[j]/255.; fixedL.cc:637-642 phi(g,n)=pow((g/255.)/4.,n-1) */
    for (size_t i = 0; i < (size_t)NT; ++i)
        for (int j = 0; j < N; ++j) {
            double g = pixels[i * (size_t)N + j] / 255.;
            double x = g / 255.;
            phi[(i * (size_t)N + j) * 2 + 0] = 1.0;            /* pow(x/4,0) */
            phi[(i * (size_t)N + j) * 2 + 1] = x / 4.;         /* pow(x/4,1) */
        }
}

orc* orc_create(int N, int NT, const double* phi, const int* labels, int nthread, int nbatch) {
    if (N < 4 || NT < 1) { fail("orc_create: need N>=4, NT>=1"); return NULL; }
    if (nthread < 1 || nthread > 16) { fail("orc_create: nthread must be 1..16 (paralleldo.h:55-56)"); return NULL; }
    if (nbatch < 1 || NT % nbatch != 0) {            /* fixedL.cc:84-89 */
        fail("totNtrain not commensurate with Nbatch"); return NULL;
    }
    orc* o = (orc*)calloc(1, sizeof(orc));
    o->N = N; o->NT = NT; o->c0 = N / 2;             /* fixedL.cc:616 */
    o->nthread = nthread; o->nbatch = nbatch; o->batchsize = NT / nbatch;   /* :90 */
    o->phi = (double*)malloc(sizeof(double) * (size_t)NT * N * 2);
    memcpy(o->phi, phi, sizeof(double) * (size_t)NT * N * 2);
    o->labels = (int*)malloc(sizeof(int) * (size_t)NT);
    memcpy(o->labels, labels, sizeof(int) * (size_t)NT);
    for (int i = 0; i < NT; ++i)
        if (labels[i] < 0 || labels[i] >= NL) { fail("label out of range"); orc_destroy(o); return NULL; }
    o->W = (site_t*)calloc((size_t)N + 2, sizeof(site_t));
    o->E = (envs_t*)calloc((size_t)N + 2, sizeof(envs_t));
    o->currb = -1;
    o->sw = 1; o->b = 1; o->ha = 1;
    return o;
}

void orc_destroy(orc* o) {
    if (!o) return;
    if (o->W) for (int j = 0; j <= o->N + 1; ++j) free(o->W[j].a);
    if (o->E) for (int j = 0; j <= o->N + 1; ++j) free(o->E[j].e);
    free(o->W); free(o->E); free(o->phi); free(o->labels); free(o->v); free(o);
}

int orc_set_site(orc* o, int j, int ml, int mr, int has_label, const double* A) {
    if (j < 1 || j > o->N) return fail("orc_set_site: site out of range");
    if ((j == o->c0) != (has_label != 0)) return fail("Label Index must sit on site N/2 only (fixedL.cc:734)");
    if (j == 1 && ml != 1) return fail("site 1 must have ml=1");
    if (j == o->N && mr != 1) return fail("site N must have mr=1");
    site_t* s = &o->W[j];
    free(s->a);
    s->ml = ml; s->mr = mr; s->L = has_label ? NL : 1;
    size_t sz = (size_t)ml * 2 * mr * s->L;
    s->a = (double*)malloc(sizeof(double) * sz);
    memcpy(s->a, A, sizeof(double) * sz);
    return 0;
}
int orc_site_dims(const orc* o, int j, int* ml, int* mr, int* has_label) {
    if (j < 1 || j > o->N || !o->W[j].a) return fail("orc_site_dims: site not set");
    *ml = o->W[j].ml; *mr = o->W[j].mr; *has_label = o->W[j].L == NL;
    return 0;
}
int orc_get_site(const orc* o, int j, double* A) {
    if (j < 1 || j > o->N || !o->W[j].a) return fail("orc_get_site: site not set");
    const site_t* s = &o->W[j];
    memcpy(A, s->a, sizeof(double) * (size_t)s->ml * 2 * s->mr * s->L);
    return 0;
}
static int check_W(const orc* o) {
    for (int j = 1; j <= o->N; ++j) {
        if (!o->W[j].a) return fail("W not fully set");
        if (j > 1 && o->W[j].ml != o->W[j - 1].mr) return fail("W bond dimensions inconsistent");
    }
    return 0;
}

/* ------------------------------------------------------------------ environments -------- */
static const double* phi_of(const orc* o, int i, int j) { return o->phi + ((size_t)i * o->N + (j - 1)) * 2; }

static void env_alloc(orc* o, int j, int m, int L) {
    envs_t* e = &o->E[j];
    if (e->m != m || e->L != L || !e->e) {
        free(e->e);
        e->m = m; e->L = L;
        e->e = (double*)malloc(sizeof(double) * (size_t)o->NT * m * L);
    }
}

/* (t.A(n)*W.A(n)) [* currE]   -- fixedL.cc:144,148 (init) and :223,227 with dir==Fromright.
   out[a(,l)] = sum_{s,beta} phi[s] A[a,s,beta(,l)] Ein[beta(,l)] ; Ein==NULL means mr==1, Ein=1 */
static void env_step_from_right(const site_t* A, const double* ph, const double* Ein, int Lin, double* out) {
    int ml = A->ml, mr = A->mr, LA = A->L, Lout = LA > Lin ? LA : Lin;
    for (int l = 0; l < Lout; ++l) {
        int la = LA == 1 ? 0 : l, li = Lin == 1 ? 0 : l;
        for (int a = 0; a < ml; ++a) {
            double acc = 0.;
            for (int be = 0; be < mr; ++be) {
                double m0 = A->a[a + (size_t)ml * (0 + 2 * (be + (size_t)mr * la))];
                double m1 = A->a[a + (size_t)ml * (1 + 2 * (be + (size_t)mr * la))];
                double e = Ein ? Ein[be + (size_t)mr * li] : 1.0;
                acc += (ph[0] * m0 + ph[1] * m1) * e;
            }
            out[a + (size_t)ml * l] = acc;
        }
    }
}
/* prevE * (t.A(c)*W.A(c))  with dir==Fromleft -- fixedL.cc:223,227.
   out[beta(,l)] = sum_{a,s} Ein[a(,l)] phi[s] A[a,s,beta(,l)] ; Ein==NULL means ml==1 */
static void env_step_from_left(const site_t* A, const double* ph, const double* Ein, int Lin, double* out) {
    int ml = A->ml, mr = A->mr, LA = A->L, Lout = LA > Lin ? LA : Lin;
    for (int l = 0; l < Lout; ++l) {
        int la = LA == 1 ? 0 : l, li = Lin == 1 ? 0 : l;
        for (int be = 0; be < mr; ++be) {
            double acc = 0.;
            for (int a = 0; a < ml; ++a) {
                double m0 = A->a[a + (size_t)ml * (0 + 2 * (be + (size_t)mr * la))];
                double m1 = A->a[a + (size_t)ml * (1 + 2 * (be + (size_t)mr * la))];
                double e = Ein ? Ein[a + (size_t)ml * li] : 1.0;
                acc += e * (ph[0] * m0 + ph[1] * m1);
            }
            out[be + (size_t)mr * l] = acc;
        }
    }
}

typedef struct { orc* o; int n; int batchStart; int from_left; int has_prev; int prev; } env_task;
static void env_task_fn(void* p, Bound b) {
    env_task* t = (env_task*)p; orc* o = t->o;
    const site_t* A = &o->W[t->n];
    envs_t* out = &o->E[t->n];
    const envs_t* in = t->has_prev ? &o->E[t->prev] : NULL;
    for (size_t k = b.begin; k < b.end; ++k) {
        int i = t->batchStart + (int)k;
        const double* Ein = in ? in->e + (size_t)i * in->m * in->L : NULL;
        double* Eo = out->e + (size_t)i * out->m * out->L;
        if (t->from_left) env_step_from_left(A, phi_of(o, i, t->n), Ein, in ? in->L : 1, Eo);
        else              env_step_from_right(A, phi_of(o, i, t->n), Ein, in ? in->L : 1, Eo);
        /* nextE.scaleTo(1.) (fixedL.cc:150,229) is value preserving: nothing to do */
    }
}

int orc_set_bond(orc* o, int b);

/* TrainStates::init -- fixedL.cc:122-157 */
int orc_init(orc* o) {
    if (check_W(o)) return -1;
    int N = o->N;
    /* The reference loops batches outermost (:133) and sites inside (:136); the result per image
       is independent of that order, envs are kept in RAM instead of proj_images/ files. */
    for (int bn = 0; bn < o->nbatch; ++bn) {
        int batchStart = bn * o->batchsize;                              /* :135 */
        for (int n = N; n >= 3; --n) {                                   /* :136 */
            const site_t* A = &o->W[n];
            int Lin = (n == N) ? 1 : o->E[n + 1].L;
            int Lout = A->L > Lin ? A->L : Lin;
            if (bn == 0) env_alloc(o, n, A->ml, Lout);
            env_task t = { o, n, batchStart, 0, n != N, n + 1 };         /* :142-149 */
            parallel_do(o->nthread, (size_t)o->batchsize, env_task_fn, &t);
        }
    }
    o->currb = -1;
    return orc_set_bond(o, 1);                                           /* :156 */
}

/* TrainStates::shiftE -- fixedL.cc:192-233 */
int orc_shiftE(orc* o, int b, int from_left) {
    int N = o->N;
    int c = from_left ? b : b + 1;                                       /* :196 */
    int prevc = from_left ? b - 1 : b + 2;                               /* :199 */
    int hasPrev = (prevc >= 1 && prevc <= N);                            /* :200 */
    const site_t* A = &o->W[c];
    int Lin = hasPrev ? o->E[prevc].L : 1;
    int Lout = A->L > Lin ? A->L : Lin;
    if (hasPrev && o->E[prevc].m != (from_left ? A->ml : A->mr)) return fail("shiftE: env/site dimension mismatch");
    env_alloc(o, c, from_left ? A->mr : A->ml, Lout);
    for (int bn = 0; bn < o->nbatch; ++bn) {                             /* :213 */
        env_task t = { o, c, bn * o->batchsize, from_left, hasPrev, prevc };
        parallel_do(o->nthread, (size_t)o->batchsize, env_task_fn, &t);  /* :217-230 */
    }
    return 0;
}

int orc_env_dims(const orc* o, int j, int* m, int* has_label) {
    if (j < 1 || j > o->N || !o->E[j].e) return fail("orc_env_dims: env not built");
    *m = o->E[j].m; *has_label = o->E[j].L == NL;
    return 0;
}
int orc_get_env(const orc* o, int j, int i, double* E) {
    if (j < 1 || j > o->N || !o->E[j].e) return fail("orc_get_env: env not built");
    const envs_t* e = &o->E[j];
    memcpy(E, e->e + (size_t)i * e->m * e->L, sizeof(double) * (size_t)e->m * e->L);
    return 0;
}

/* ------------------------------------------------------------------ setBond ------------- */
typedef struct { orc* o; int b; int batchStart; int useL, useR; } sb_task;
static void sb_task_fn(void* p, Bound bd) {
    sb_task* t = (sb_task*)p; orc* o = t->o;
    int lc = t->b - 1, rc = t->b + 2;
    int mL = o->vmL, mR = o->vmR, vL = o->vL;
    const envs_t* LE = t->useL ? &o->E[lc] : NULL;
    const envs_t* RE = t->useR ? &o->E[rc] : NULL;
    for (size_t k = bd.begin; k < bd.end; ++k) {
        int i = t->batchStart + (int)k;
        const double* pa = phi_of(o, i, lc + 1);
        const double* pb = phi_of(o, i, rc - 1);
        const double* le = LE ? LE->e + (size_t)i * LE->m * LE->L : NULL;
        const double* re = RE ? RE->e + (size_t)i * RE->m * RE->L : NULL;
        double* v = o->v + (size_t)i * o->vsz;
        /* t.v = t.A(lc+1)*t.A(rc-1); if(useL) t.v *= LE; if(useR) t.v *= RE;  (fixedL.cc:183-185)
           -- a pure outer product: v[a,s,t,beta(,l)] */
        for (int l = 0; l < vL; ++l)
            for (int be = 0; be < mR; ++be) {
                double r = re ? re[be + (size_t)mR * (RE->L == 1 ? 0 : l)] : 1.0;
                for (int tt = 0; tt < 2; ++tt)
                    for (int s = 0; s < 2; ++s)
                        for (int a = 0; a < mL; ++a) {
                            double lv = le ? le[a + (size_t)mL * (LE->L == 1 ? 0 : l)] : 1.0;
                            v[a + (size_t)mL * (s + 2 * (tt + 2 * (be + (size_t)mR * l)))] = pa[s] * pb[tt] * lv * r;
                        }
            }
    }
}

/* TrainStates::setBond -- fixedL.cc:159-190 */
int orc_set_bond(orc* o, int b) {
    if (b < 1 || b > o->N - 1) return fail("orc_set_bond: bond out of range");
    if (o->currb == b) return 0;                                         /* :162 */
    o->currb = b;
    int lc = b - 1, rc = b + 2;                                          /* :164-165 */
    int useL = lc > 0, useR = rc < o->N + 1;                             /* :166-167 */
    if (useL && !o->E[lc].e) return fail("setBond: left env missing");
    if (useR && !o->E[rc].e) return fail("setBond: right env missing");
    int mL = useL ? o->E[lc].m : 1, mR = useR ? o->E[rc].m : 1;
    int vL = ((useL && o->E[lc].L == NL) || (useR && o->E[rc].L == NL)) ? NL : 1;
    if (mL != o->W[b].ml || mR != o->W[b + 1].mr) return fail("setBond: env dims do not match W");
    size_t vsz = (size_t)mL * 4 * mR * vL;
    if (vsz != o->vsz || !o->v) {
        free(o->v);
        o->v = (double*)malloc(sizeof(double) * vsz * (size_t)o->NT);
        if (!o->v) return fail("setBond: out of memory for dense t.v");
    }
    o->vsz = vsz; o->vmL = mL; o->vmR = mR; o->vL = vL;
    for (int bn = 0; bn < o->nbatch; ++bn) {                             /* :174 */
        sb_task t = { o, b, bn * o->batchsize, useL, useR };
        parallel_do(o->nthread, (size_t)o->batchsize, sb_task_fn, &t);   /* :179-186 */
    }
    return 0;
}

/* ------------------------------------------------------------------ bond tensor --------- */
int orc_bond_dims(const orc* o, int b, int* mL, int* mR, int* label_on_B) {
    if (b < 1 || b > o->N - 1 || !o->W[b].a || !o->W[b + 1].a) return fail("orc_bond_dims: bad bond");
    *mL = o->W[b].ml; *mR = o->W[b + 1].mr;
    *label_on_B = (o->c0 == b || o->c0 == b + 1);
    return 0;
}
/* oB = W.A(c)*W.A(c+dc) -- fixedL.cc:494 (also :527, :745): contraction over the shared link */
int orc_bond_tensor(const orc* o, int b, double* B) {
    int mL, mR, lab;
    if (orc_bond_dims(o, b, &mL, &mR, &lab)) return -1;
    const site_t* A1 = &o->W[b]; const site_t* A2 = &o->W[b + 1];
    int k = A1->mr, LB = lab ? NL : 1;
    for (int l = 0; l < LB; ++l) {
        int l1 = A1->L == 1 ? 0 : l, l2 = A2->L == 1 ? 0 : l;
        for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t) for (int s = 0; s < 2; ++s) for (int a = 0; a < mL; ++a) {
            double acc = 0.;
            for (int g = 0; g < k; ++g)
                acc += A1->a[a + (size_t)mL * (s + 2 * (g + (size_t)k * l1))] * A2->a[g + (size_t)k * (t + 2 * (be + (size_t)mR * l2))];
            B[a + (size_t)mL * (s + 2 * (t + 2 * (be + (size_t)mR * l)))] = acc;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ per-image algebra --- */
/* P = B*t.v  (fixedL.cc:318,377,399,416): contraction over all shared indices, Label left open */
static void image_forward(const orc* o, const double* B, int i, double* P) {
    size_t sz = (size_t)o->vmL * 4 * o->vmR;
    const double* v = o->v + (size_t)i * o->vsz;
    if (o->vL == NL) {            /* Label lives on t.v (one of the envs) */
        for (int l = 0; l < NL; ++l) {
            const double* vl = v + sz * l; double acc = 0.;
            for (size_t x = 0; x < sz; ++x) acc += B[x] * vl[x];
            P[l] = acc;
        }
    } else {                      /* Label lives on B (c0 in {b,b+1}) */
        for (int l = 0; l < NL; ++l) {
            const double* Bl = B + sz * l; double acc = 0.;
            for (size_t x = 0; x < sz; ++x) acc += Bl[x] * v[x];
            P[l] = acc;
        }
    }
}
/* T += dP*dag(t.v)  (fixedL.cc:379,418) */
static void image_backward(const orc* o, const double* dP, int i, double* T) {
    size_t sz = (size_t)o->vmL * 4 * o->vmR;
    const double* v = o->v + (size_t)i * o->vsz;
    if (o->vL == NL) {
        for (int l = 0; l < NL; ++l) {
            const double* vl = v + sz * l; double d = dP[l];
            for (size_t x = 0; x < sz; ++x) T[x] += d * vl[x];
        }
    } else {
        for (int l = 0; l < NL; ++l) {
            double* Tl = T + sz * l; double d = dP[l];
            for (size_t x = 0; x < sz; ++x) Tl[x] += d * v[x];
        }
    }
}
static size_t bond_size(const orc* o) { return (size_t)o->vmL * 4 * o->vmR * (o->vL == NL ? 1 : NL); }
static double sqnorm(const double* x, size_t n) { double s = 0.; for (size_t k = 0; k < n; ++k) s += x[k] * x[k]; return s; }

/* TrainStates::execute -- fixedL.cc:236-253: batches sequential, chunks fork-joined, each chunk
   walks its images in order and calls f(thread_id, t) */
typedef void (*img_fn)(void* ctx, int nt, int i);
typedef struct { img_fn f; void* ctx; int batchStart; } ex_task;
static void ex_task_fn(void* p, Bound b) {
    ex_task* t = (ex_task*)p;
    for (size_t i = t->batchStart + b.begin; i < t->batchStart + b.end; ++i) t->f(t->ctx, (int)b.n, (int)i);
}
static void ts_execute(const orc* o, img_fn f, void* ctx) {
    for (int bn = 0; bn < o->nbatch; ++bn) {
        ex_task t = { f, ctx, bn * o->batchsize };
        parallel_do(o->nthread, (size_t)o->batchsize, ex_task_fn, &t);
    }
}

int orc_forward(const orc* o, const double* B, double* P) {
    if (!o->v) return fail("orc_forward: setBond not called");
    for (int i = 0; i < o->NT; ++i) image_forward(o, B, i, P + (size_t)i * NL);
    return 0;
}

typedef struct { const orc* o; const double* B; double* tensors; size_t bsz; double* reals; int with_cost; } grad_ctx;
static void grad_img(void* p, int nt, int i) {
    grad_ctx* c = (grad_ctx*)p;
    double P[NL], dP[NL];
    image_forward(c->o, c->B, i, P);                                    /* :377 / :416 */
    for (int l = 0; l < NL; ++l) dP[l] = (l == c->o->labels[i] ? 1.0 : 0.0) - P[l];   /* :378 / :417 */
    image_backward(c->o, dP, i, c->tensors + (size_t)nt * c->bsz);      /* :379 / :418 */
    if (c->with_cost) c->reals[nt] += sqnorm(dP, NL);                   /* :419 sqr(norm(dP)) */
}
/* gradient evaluation shared by fixedL.cc:374-385 and :412-421; out = accumulate(tensors) in
   thread order (:385,:421); *csum = accumulate(reals) (:427) when with_cost */
static void eval_gradient(const orc* o, const double* B, double* out, double* csum) {
    size_t bsz = bond_size(o);
    int nth = o->nthread;
    double* tensors = (double*)calloc(bsz * (size_t)nth, sizeof(double));
    double* reals = (double*)calloc((size_t)nth, sizeof(double));
    grad_ctx c = { o, B, tensors, bsz, reals, csum != NULL };
    ts_execute(o, grad_img, &c);
    memset(out, 0, sizeof(double) * bsz);
    for (int n = 0; n < nth; ++n) for (size_t x = 0; x < bsz; ++x) out[x] += tensors[(size_t)n * bsz + x];
    if (csum) { double s = 0.; for (int n = 0; n < nth; ++n) s += reals[n]; *csum = s; }
    free(tensors); free(reals);
}
int orc_gradient(const orc* o, const double* B, double* G) {
    if (!o->v) return fail("orc_gradient: setBond not called");
    eval_gradient(o, B, G, NULL);
    return 0;
}

/* util.h:42-57 argmax: first maximum */
static int argmax10(const double* c) {
    double mel = c[0]; int mn = 0;
    for (int n = 0; n < NL; ++n) if (c[n] > mel) { mel = c[n]; mn = n; }
    return mn;
}

typedef struct { const orc* o; const double* B; double* reals; /* [10][nthread] */ int* ints; } qc_ctx;
static void qc_img(void* p, int nt, int i) {
    qc_ctx* c = (qc_ctx*)p; const orc* o = c->o;
    double P[NL], dP[NL], weights[NL];
    image_forward(o, c->B, i, P);                                       /* :318 */
    int tl = o->labels[i];
    for (int l = 0; l < NL; ++l) dP[l] = (l == tl ? 1.0 : 0.0) - P[l];  /* :319 */
    c->reals[(size_t)tl * o->nthread + nt] += sqnorm(dP, NL);           /* :320 */
    for (int l = 0; l < NL; ++l) weights[l] = fabs(P[l]);               /* :321-324 */
    if (tl == argmax10(weights)) c->ints[nt] += 1;                      /* :326 */
}
/* quadcost -- fixedL.cc:280-344 ("Normalize" is always false on this path, :467) */
double orc_quadcost(const orc* o, const double* B, double lambda, double label_cost[NL], double* reg_cost, long* ncorrect) {
    int nth = o->nthread;
    double* reals = (double*)calloc((size_t)NL * nth, sizeof(double));
    int* ints = (int*)calloc((size_t)nth, sizeof(int));
    qc_ctx c = { o, B, reals, ints };
    ts_execute(o, qc_img, &c);
    double CR = lambda * sqnorm(B, bond_size(o));                       /* :329 */
    double C = 0.;
    for (int l = 0; l < NL; ++l) {                                      /* :331-336 */
        double CL = 0.; for (int n = 0; n < nth; ++n) CL += reals[(size_t)l * nth + n];
        if (label_cost) label_cost[l] = CL;
        C += CL;
    }
    C += CR;                                                            /* :338 */
    long ncor = 0; for (int n = 0; n < nth; ++n) ncor += ints[n];       /* :339 */
    if (reg_cost) *reg_cost = CR;
    if (ncorrect) *ncorrect = ncor;
    free(reals); free(ints);
    return C;
}

typedef struct { const orc* o; const double* p; double* reals; } pap_ctx;
static void pap_img(void* q, int nt, int i) {
    pap_ctx* c = (pap_ctx*)q;
    double pv[NL];
    image_forward(c->o, c->p, i, pv);                                   /* :399 */
    c->reals[nt] += sqnorm(pv, NL);                                     /* :400 */
}

/* cgrad -- fixedL.cc:349-445 */
int orc_cgrad(const orc* o, double* B, int npass, double lambda, double cconv, orc_cg_trace* tr) {
    if (!o->v) return fail("orc_cgrad: setBond not called");
    if (npass > 64) return fail("orc_cgrad: npass > 64");
    size_t bsz = bond_size(o);
    int nth = o->nthread;
    double* r = (double*)malloc(sizeof(double) * bsz);
    double* p = (double*)malloc(sizeof(double) * bsz);
    double* nr = (double*)malloc(sizeof(double) * bsz);
    double* reals = (double*)malloc(sizeof(double) * (size_t)nth);
    if (tr) memset(tr, 0, sizeof *tr);

    eval_gradient(o, B, r, NULL);                                       /* :374-385 */
    if (lambda != 0.) for (size_t x = 0; x < bsz; ++x) r[x] = r[x] - lambda * B[x];   /* :386 */
    memcpy(p, r, sizeof(double) * bsz);                                 /* :388 */
    for (int pass = 1; pass <= npass; ++pass) {                         /* :389 */
        for (int n = 0; n < nth; ++n) reals[n] = 0.;                    /* :393 */
        pap_ctx pc = { o, p, reals };
        ts_execute(o, pap_img, &pc);                                    /* :394-401 */
        double pAp = 0.; for (int n = 0; n < nth; ++n) pAp += reals[n]; /* :402 */
        pAp += lambda * sqnorm(p, bsz);                                 /* :403 */
        double a = sqnorm(r, bsz) / pAp;                                /* :405 */
        for (size_t x = 0; x < bsz; ++x) B[x] = B[x] + a * p[x];        /* :406 */
        if (tr) { tr->npass_done = pass; tr->pAp[pass - 1] = pAp; tr->alpha[pass - 1] = a; }
        if (pass == npass) break;                                       /* :409 */

        double csum = 0.;
        eval_gradient(o, B, nr, &csum);                                 /* :412-421 */
        if (lambda != 0.) for (size_t x = 0; x < bsz; ++x) nr[x] = nr[x] - lambda * B[x];   /* :422 */
        double q = sqrt(sqnorm(nr, bsz)) / sqrt(sqnorm(r, bsz));
        double beta = q * q;                                            /* :423 sqr(norm(nr)/norm(r)) */
        memcpy(r, nr, sizeof(double) * bsz);                            /* :424 */
        double C = csum + lambda * sqnorm(B, bsz);                      /* :427-428 */
        double rn = sqrt(sqnorm(r, bsz));
        if (tr) { tr->cost[pass - 1] = C; tr->rnorm[pass - 1] = rn; }
        if (rn < cconv) { if (tr) tr->converged = 1; break; }           /* :432-436 */
        for (size_t x = 0; x < bsz; ++x) p[x] = r[x] + beta * p[x];     /* :442 */
    }
    free(r); free(p); free(nr); free(reals);
    return 0;
}

/* ------------------------------------------------------------------ SVD + truncation ---- */
/* One-sided Jacobi (Hestenes) SVD of the R x C column-major matrix M (R >= C): on return the
   columns of M are U*diag(s) un-normalised -> we normalise; V is C x C.  Stand-in for ITensor's
   SVD; same singular values / vectors up to sign. */
static void jacobi_svd_tall(int R, int C, double* M, double* s, double* V) {
    for (int j = 0; j < C; ++j) for (int i = 0; i < C; ++i) V[i + (size_t)C * j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < C - 1; ++p) for (int q = p + 1; q < C; ++q) {
            double* mp = M + (size_t)R * p; double* mq = M + (size_t)R * q;
            double alpha = 0., beta = 0., gamma = 0.;
            for (int i = 0; i < R; ++i) { alpha += mp[i] * mp[i]; beta += mq[i] * mq[i]; gamma += mp[i] * mq[i]; }
            if (gamma == 0. || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
            rotated = 1;
            double zeta = (beta - alpha) / (2. * gamma);
            double t = (zeta >= 0. ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
            double c = 1. / sqrt(1. + t * t), sn = c * t;
            for (int i = 0; i < R; ++i) { double x = mp[i], y = mq[i]; mp[i] = c * x - sn * y; mq[i] = sn * x + c * y; }
            double* vp = V + (size_t)C * p; double* vq = V + (size_t)C * q;
            for (int i = 0; i < C; ++i) { double x = vp[i], y = vq[i]; vp[i] = c * x - sn * y; vq[i] = sn * x + c * y; }
        }
        if (!rotated) break;
    }
    for (int j = 0; j < C; ++j) {
        double n2 = 0.; double* mj = M + (size_t)R * j;
        for (int i = 0; i < R; ++i) n2 += mj[i] * mj[i];
        s[j] = sqrt(n2);
        if (s[j] > 0.) for (int i = 0; i < R; ++i) mj[i] /= s[j];
    }
}

/* thin SVD of R x C column-major A (any shape): U R x k, s[k] descending, Vt k x C, k=min(R,C) */
void orc_thin_svd(int R, int C, const double* A, double* U, double* s, double* Vt) {
    int k = R < C ? R : C;
    int tall = R >= C;
    int r = tall ? R : C, c = tall ? C : R;       /* work on the tall orientation */
    double* M = (double*)malloc(sizeof(double) * (size_t)r * c);
    double* V = (double*)malloc(sizeof(double) * (size_t)c * c);
    double* sv = (double*)malloc(sizeof(double) * (size_t)c);
    if (tall) memcpy(M, A, sizeof(double) * (size_t)R * C);
    else for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) M[j + (size_t)C * i] = A[i + (size_t)R * j];
    jacobi_svd_tall(r, c, M, sv, V);
    int* ord = (int*)malloc(sizeof(int) * (size_t)c);
    for (int j = 0; j < c; ++j) ord[j] = j;
    for (int i = 1; i < c; ++i) {                 /* insertion sort, descending, stable */
        int x = ord[i], j = i - 1;
        while (j >= 0 && sv[ord[j]] < sv[x]) { ord[j + 1] = ord[j]; --j; }
        ord[j + 1] = x;
    }
    for (int g = 0; g < k; ++g) {
        int j = ord[g];
        s[g] = sv[j];
        if (tall) {   /* A = (M) diag(s) V^T */
            for (int i = 0; i < R; ++i) U[i + (size_t)R * g] = M[i + (size_t)r * j];
            for (int i = 0; i < C; ++i) Vt[g + (size_t)k * i] = V[i + (size_t)c * j];
        } else {      /* A^T = M diag(s) V^T  ->  A = V diag(s) M^T */
            for (int i = 0; i < R; ++i) U[i + (size_t)R * g] = V[i + (size_t)c * j];
            for (int i = 0; i < C; ++i) Vt[g + (size_t)k * i] = M[i + (size_t)r * j];
        }
    }
    free(M); free(V); free(sv); free(ord);
}

/* ITensor v2 truncate() as recalled in SURVEY.md 8(a9) [ITensor-recall]: P = sigma^2 sorted
   descending; always cut down to maxm; then, with scale = sum(P) (DoRelCutoff default true for
   svd), keep discarding the smallest while (discarded + P_n) < cutoff*scale and n >= minm
   (0-based n, i.e. kept > minm); truncerr = discarded/scale. */
int orc_truncate(const double* P, int origm, int maxm, int minm, double cutoff, double* truncerr) {
    if (origm == 1) { if (truncerr) *truncerr = 0.; return 1; }
    int n = origm - 1;
    double te = 0.;
    while (n >= maxm) { te += P[n]; --n; }
    double scale = 0.; for (int j = 0; j < origm; ++j) scale += P[j];
    if (scale == 0.) scale = 1.;
    while (n >= 0 && te + P[n] < cutoff * scale && n >= minm) { te += P[n]; --n; }
    if (n < 0) n = 0;
    if (truncerr) *truncerr = te / scale;
    return n + 1;
}

/* svd(B, W.Aref(c), S, W.Aref(c+dc), svd_args); W.Aref(c+dc) *= S -- fixedL.cc:519-521.
   Rows of the matrix = the indices B shares with the incoming W.A(c): outer link of c, site c and
   Label iff it lives on c. */
int orc_svd_split(orc* o, const double* B, int b, int ha, double cutoff, int maxm, int minm,
                  double* truncerr, int* newm, double* sv_out, int* nsv) {
    int mL, mR, lab;
    if (orc_bond_dims(o, b, &mL, &mR, &lab)) return -1;
    int LB = lab ? NL : 1;
    int labL = (o->c0 == b), labR = (o->c0 == b + 1);
    int nl = 2 * mL * (labL ? NL : 1);      /* (a,s[,l]) */
    int nr = 2 * mR * (labR ? NL : 1);      /* (t,beta[,l]) */
    int R = ha == 1 ? nl : nr, C = ha == 1 ? nr : nl;
    double* M = (double*)malloc(sizeof(double) * (size_t)R * C);
    for (int l = 0; l < LB; ++l) for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 2; ++s) for (int a = 0; a < mL; ++a) {
            int il = a + mL * s + (labL ? 2 * mL * l : 0);
            int ir = t + 2 * be + (labR ? 2 * mR * l : 0);
            double x = B[a + (size_t)mL * (s + 2 * (t + 2 * (be + (size_t)mR * l)))];
            if (ha == 1) M[il + (size_t)R * ir] = x; else M[ir + (size_t)R * il] = x;
        }
    int k = R < C ? R : C;
    double* U = (double*)malloc(sizeof(double) * (size_t)R * k);
    double* s = (double*)malloc(sizeof(double) * (size_t)k);
    double* Vt = (double*)malloc(sizeof(double) * (size_t)k * C);
    orc_thin_svd(R, C, M, U, s, Vt);
    double* P = (double*)malloc(sizeof(double) * (size_t)k);
    for (int g = 0; g < k; ++g) P[g] = s[g] * s[g];
    double te = 0.;
    int m = orc_truncate(P, k, maxm, minm, cutoff, &te);
    if (truncerr) *truncerr = te;
    if (newm) *newm = m;
    if (nsv) *nsv = k;
    if (sv_out) memcpy(sv_out, s, sizeof(double) * (size_t)k);

    site_t* Sl = &o->W[b]; site_t* Sr = &o->W[b + 1];
    free(Sl->a); free(Sr->a);
    Sl->ml = mL; Sl->mr = m; Sl->L = labL ? NL : 1;
    Sr->ml = m; Sr->mr = mR; Sr->L = labR ? NL : 1;
    Sl->a = (double*)malloc(sizeof(double) * (size_t)mL * 2 * m * Sl->L);
    Sr->a = (double*)malloc(sizeof(double) * (size_t)m * 2 * mR * Sr->L);
    for (int g = 0; g < m; ++g) {
        /* left site  A_b[a,s,g(,l)]   : U (ha==1) or S*V (ha==2) */
        for (int l = 0; l < Sl->L; ++l) for (int sI = 0; sI < 2; ++sI) for (int a = 0; a < mL; ++a) {
            int il = a + mL * sI + (labL ? 2 * mL * l : 0);
            double x = ha == 1 ? U[il + (size_t)R * g] : s[g] * Vt[g + (size_t)k * il];
            Sl->a[a + (size_t)mL * (sI + 2 * (g + (size_t)m * l))] = x;
        }
        /* right site A_{b+1}[g,t,beta(,l)] : S*V (ha==1) or U (ha==2) */
        for (int l = 0; l < Sr->L; ++l) for (int be = 0; be < mR; ++be) for (int t = 0; t < 2; ++t) {
            int ir = t + 2 * be + (labR ? 2 * mR * l : 0);
            double x = ha == 1 ? s[g] * Vt[g + (size_t)k * ir] : U[ir + (size_t)R * g];
            Sr->a[g + (size_t)m * (t + 2 * (be + (size_t)mR * l))] = x;
        }
    }
    free(M); free(U); free(s); free(Vt); free(P);
    return 0;
}

/* ITensor sweepnext(b,ha,N) as recalled in SURVEY.md 8(a12) [ITensor-recall] */
void orc_sweepnext(int* b, int* ha, int N) {
    int inc = (*ha == 1) ? +1 : -1;
    *b += inc;
    if (*b == ((*ha == 1) ? N : 0)) { *b -= inc; ++*ha; }
}

/* ------------------------------------------------------------------ mldmrg -------------- */
/* fixedL.cc:451-570 */
int orc_mldmrg(orc* o, int nsweep, int maxm, int minm, double cutoff, int npass, double lambda,
               double cconv, int max_bonds, orc_bond_report* reports, int verbose) {
    int N = o->N, done = 0;
    double NT = (double)o->NT;
    while (o->sw <= nsweep) {                                            /* :470 */
        if (max_bonds > 0 && done >= max_bonds) break;
        int b = o->b, ha = o->ha, sw = o->sw;
        int c = (ha == 1) ? b : b + 1;                                   /* :482 */
        int dc = (ha == 1) ? +1 : -1;                                    /* :483 */
        if (orc_set_bond(o, b)) return -1;                               /* :488 */
        if (verbose) printf("Sweep %d Half %d Bond %d\n", sw, ha, c);    /* :490 */
        int mL, mR, lab;
        orc_bond_dims(o, b, &mL, &mR, &lab);
        int origm = o->W[b].mr;                                          /* :493 */
        size_t bsz = (size_t)mL * 4 * mR * (lab ? NL : 1);
        double* B = (double*)malloc(sizeof(double) * bsz);
        double* newB = (double*)malloc(sizeof(double) * bsz);
        orc_bond_tensor(o, b, B);                                        /* :494-498 */
        orc_bond_report rep; memset(&rep, 0, sizeof rep);
        rep.sweep = sw; rep.half = ha; rep.bond = b; rep.c = c; rep.origm = origm;
        if (verbose) printf("In cgrad, lambda = %.3E\n", lambda);        /* :358 */
        if (orc_cgrad(o, B, npass, lambda, cconv, &rep.cg)) return -1;   /* :504 */
        if (verbose) for (int p = 0; p < rep.cg.npass_done; ++p) {
            printf("  Conj grad pass %d\n", p + 1);
            if (p + 1 < npass) { printf("  Cost = %.10f\n", rep.cg.cost[p] / NT); printf("  |r| = %.1E\n", rep.cg.rnorm[p]); }
        }
        if (orc_svd_split(o, B, b, ha, cutoff, maxm, minm, &rep.truncerr, &rep.newm, NULL, NULL)) return -1;   /* :519-522 */
        orc_bond_tensor(o, b, newB);                                     /* :527 */
        rep.norm_newB = sqrt(sqnorm(newB, bsz));                         /* :528 */
        double d2 = 0.; for (size_t x = 0; x < bsz; ++x) { double d = B[x] - newB[x]; d2 += d * d; }
        rep.diff_B_newB = sqrt(d2);                                      /* :530 */
        /* cargs keeps the lambda captured at :467; no LAMBDA reload here */
        rep.cost_after_svd = orc_quadcost(o, newB, lambda, rep.label_cost, &rep.reg_cost, &rep.ncorrect);   /* :532 */
        if (verbose) {
            printf("SVD trunc err = %.2E\n", rep.truncerr);              /* :523 */
            printf("Original m=%d, New m=%d\n", origm, rep.newm);        /* :525 */
            printf("|B-newB| = %.3E\n", rep.diff_B_newB);
            printf("Percent correct = %.4f%%, # incorrect = %ld/%d\n", rep.ncorrect * 100. / NT, (long)o->NT - rep.ncorrect, o->NT);
            printf("--> After SVD, Cost = %.10f\n", rep.cost_after_svd / NT);   /* :533 */
        }
        if (orc_shiftE(o, b, ha == 1)) return -1;                        /* :540 */
        free(B); free(newB);
        if (reports) reports[done] = rep;
        ++done;
        (void)dc;
        orc_sweepnext(&b, &ha, N);                                       /* :478 */
        if (ha > 2) { b = 1; ha = 1; o->sw = sw + 1; }                   /* loop ends at ha==3 -> next sweep */
        o->b = b; o->ha = ha;
    }
    return done;
}

/* util.h:19-40 toverlap(psi,img,c) with c = the Label site (util.h:129-140 finds it) */
int orc_toverlap(const orc* o, int i, double* out) {
    if (check_W(o)) return -1;
    int N = o->N, c = o->c0;
    int cap = 1; for (int j = 1; j <= N; ++j) { if (o->W[j].ml > cap) cap = o->W[j].ml; if (o->W[j].mr > cap) cap = o->W[j].mr; }
    double* cur = (double*)malloc(sizeof(double) * (size_t)cap * NL);
    double* nxt = (double*)malloc(sizeof(double) * (size_t)cap * NL);
    double* left = (double*)malloc(sizeof(double) * (size_t)cap * NL);
    /* W = img.A(N)*psi.A(N); for j=N-1..c: W *= img.A(j)*psi.A(j)   (util.h:24-29) */
    int L = 1;
    env_step_from_right(&o->W[N], phi_of(o, i, N), NULL, 1, cur);
    for (int j = N - 1; j >= c; --j) {
        env_step_from_right(&o->W[j], phi_of(o, i, j), cur, L, nxt);
        if (o->W[j].L == NL) L = NL;
        double* t = cur; cur = nxt; nxt = t;
    }
    /* cur: [ml_c][10] */
    if (c > 1) {                                                         /* util.h:30-38 */
        env_step_from_left(&o->W[1], phi_of(o, i, 1), NULL, 1, left);
        for (int j = 2; j < c; ++j) {
            env_step_from_left(&o->W[j], phi_of(o, i, j), left, 1, nxt);
            memcpy(left, nxt, sizeof(double) * (size_t)o->W[j].mr);
        }
        int m = o->W[c].ml;
        for (int l = 0; l < NL; ++l) { double acc = 0.; for (int a = 0; a < m; ++a) acc += cur[a + (size_t)m * l] * left[a]; out[l] = acc; }
    } else {
        for (int l = 0; l < NL; ++l) out[l] = cur[l];
    }
    free(cur); free(nxt); free(left);
    return 0;
}
