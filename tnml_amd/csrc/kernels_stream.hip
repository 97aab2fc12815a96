// kernels_stream.hip -- HBM-bound streaming kernels of the bond contraction (gfx950).
//
// k_labeldot : P[l][n] = sum_q A[l][q][n] * Bv[q][n]  -- the contraction of the feature-GEMM
//              output with the Label-carrying environment (second half of B*t.v,
//              fixedL.cc:318,377,399,416), fused with dP = delta_{l_n} - P (:319,378,417), the
//              per-label cost partials (:320,419), the argmax|P| count (:321-326, util.h:42-57)
//              or |P|^2 for pAp (:400).  Image-fastest layout: every lane owns 2 images, every
//              load is an 8-byte-per-lane fully coalesced stream; no cross-lane reduction.
// k_zprime   : Z'[q][n] = sum_l EL[l][q][n] * dP[l][n]  (first half of dP*dag(t.v), :379,418)
// k_features : TState ctor (fixedL.cc:28-47) with phi of :637-642 from raw bytes.
#include "tnml_internal.h"


template <typename T> struct vec2;
template <> struct vec2<float> { typedef float2 type; };
template <> struct vec2<double> { typedef double2 type; };
// non-temporal 2-element load (data streamed once per launch)
static __device__ __forceinline__ double2 load2_nt(const double* p) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(t.x, t.y);
}
static __device__ __forceinline__ float2 load2_nt(const float* p) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
    return make_float2(t.x, t.y);
}

// TA: element type of the label-carrying operand, TB: of the label-free one, TC: arithmetic type
// IPL: images per lane.  2 (128 images per workgroup, 16-byte loads) is the streaming configuration; 1 with more
// waves per workgroup keeps enough loads in flight when a rank holds few images (multi-GPU shards, small sets).
// NLT: label extent (10, or 1 in TNML_MODE_SINGLE) -- a template parameter so that the label loops stay straight-line code
// (a run-time bound costs ~35 % of this kernel's bandwidth: the loads can no longer be hoisted ahead of the FMAs).
// QU: rows of the contraction index fetched per loop iteration and wave (QU = 2: twice the bytes in flight per wave -- what a
// launch over HALF the images needs to pull the same bandwidth out of half the workgroups).
template <int NW, int IPL, int NLT, typename TA, typename TB, typename TC, int QU = 1>
__global__ __launch_bounds__(64 * NW) void k_labeldot(LdotArgs A, double* __restrict__ partials) {
    constexpr int LDI = 64 * IPL;
    __shared__ __attribute__((aligned(16))) TC red[NW * NLT * LDI];
    __shared__ double s_part[(LDI / 64) * 12];
    typedef typename vec2<TA>::type TA2;
    typedef typename vec2<TB>::type TB2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int NTp = A.NTp;
    const int blk = blockIdx.x + A.blk_off;
    const int n = blk * LDI + lane * IPL;
    const TA* Ap = static_cast<const TA*>(A.A);
    const TB* Bp = static_cast<const TB*>(A.Bv);

    TC px[NLT], py[NLT];
#pragma unroll
    for (int l = 0; l < NLT; ++l) { px[l] = 0; py[l] = 0; }
    if constexpr (QU == 2 && IPL == 2) {
        for (int q = 2 * w; q < A.mq; q += 2 * NW) {
            const bool two = q + 1 < A.mq;
            const TA* ap = Ap + (size_t)q * NTp + n;
            const TB2 u0 = *reinterpret_cast<const TB2*>(Bp + (size_t)q * NTp + n);
            TB2 u1; u1.x = 0; u1.y = 0;
            if (two) u1 = *reinterpret_cast<const TB2*>(Bp + (size_t)(q + 1) * NTp + n);
            TA2 e0[NLT], e1[NLT];
#pragma unroll
            for (int l = 0; l < NLT; ++l) {
                e0[l] = A.nt ? load2_nt(ap + (size_t)l * A.A_lstride) : *reinterpret_cast<const TA2*>(ap + (size_t)l * A.A_lstride);
                if (two) e1[l] = A.nt ? load2_nt(ap + (size_t)l * A.A_lstride + NTp) : *reinterpret_cast<const TA2*>(ap + (size_t)l * A.A_lstride + NTp);
                else { e1[l].x = 0; e1[l].y = 0; }
            }
#pragma unroll
            for (int l = 0; l < NLT; ++l) {                         // same order of accumulation as QU = 1: q, then q + 1
                px[l] = fma((TC)e0[l].x, (TC)u0.x, px[l]);
                py[l] = fma((TC)e0[l].y, (TC)u0.y, py[l]);
                if (two) { px[l] = fma((TC)e1[l].x, (TC)u1.x, px[l]); py[l] = fma((TC)e1[l].y, (TC)u1.y, py[l]); }
            }
        }
    } else
    for (int q = w; q < A.mq; q += NW) {
        const TA* ap = Ap + (size_t)q * NTp + n;
        if constexpr (IPL == 2) {
            const TB2 u = *reinterpret_cast<const TB2*>(Bp + (size_t)q * NTp + n);
#pragma unroll
            for (int l = 0; l < NLT; ++l) {
                const TA2 e = A.nt ? load2_nt(ap + (size_t)l * A.A_lstride) : *reinterpret_cast<const TA2*>(ap + (size_t)l * A.A_lstride);
                px[l] = fma((TC)e.x, (TC)u.x, px[l]);
                py[l] = fma((TC)e.y, (TC)u.y, py[l]);
            }
        } else {
            const TC u = (TC)Bp[(size_t)q * NTp + n];
#pragma unroll
            for (int l = 0; l < NLT; ++l) px[l] = fma((TC)ap[(size_t)l * A.A_lstride], u, px[l]);
        }
    }
#pragma unroll
    for (int l = 0; l < NLT; ++l) {
        red[(w * NLT + l) * LDI + lane * IPL] = px[l];
        if constexpr (IPL == 2) red[(w * NLT + l) * LDI + lane * 2 + 1] = py[l];
    }
    __syncthreads();

    if (tid < LDI) {
        const int ni = blk * LDI + tid;
        TC P[NLT];
#pragma unroll
        for (int l = 0; l < NLT; ++l) {
            TC s = 0;
            for (int ww = 0; ww < NW; ++ww) s += red[(ww * NLT + l) * LDI + tid];   // fixed order
            P[l] = s;
        }
        const int lab = A.label[ni];
        TC val = 0; int cor = 0;
        TC* Pout = static_cast<TC*>(A.P);
        TC* dPout = static_cast<TC*>(A.dP);
        if (A.mode == LD_MODE_PAP) {
#pragma unroll
            for (int l = 0; l < NLT; ++l) {
                val = fma(P[l], P[l], val);                                            // sqr(norm(pv)), :400
                if (Pout) Pout[(size_t)l * NTp + ni] = P[l];
            }
            if (lab < 0) val = 0;
        } else {
            TC best = fabs(P[0]); int arg = 0;
#pragma unroll
            for (int l = 0; l < NLT; ++l) {
                const TC tgt = A.target < 0 ? (l == lab ? (TC)1 : (TC)0) : (lab == A.target ? (TC)1 : (TC)0);   // single.h:103,193
                const TC d = (lab >= 0) ? (tgt - P[l]) : (TC)0;                        // deltas[t.l] - P
                val = fma(d, d, val);
                if (dPout) dPout[(size_t)l * NTp + ni] = d;
                if (Pout) Pout[(size_t)l * NTp + ni] = P[l];
                const TC wgt = fabs(P[l]);
                if (wgt > best) { best = wgt; arg = l; }                               // first maximum
            }
            if (A.target < 0) cor = (lab >= 0 && arg == lab) ? 1 : 0;
            else              cor = (lab >= 0 && ((P[0] > (TC)0.5) == (lab == A.target))) ? 1 : 0;
        }
        wave_bucket_partials((double)val, lab, cor, A.mode == LD_MODE_PAP, s_part, tid >> 6, tid & 63);
    }
    __syncthreads();
    sum_wave_partials(s_part, LDI / 64, partials + (size_t)blk * 12, tid);
}

__global__ __launch_bounds__(768) void k_reduce_partials(const double* __restrict__ partials, int nblk, double* __restrict__ out, int only_sum) {
    // 12 waves, wave t sums column t: lane i takes rows i, i+64, ... in order, then a fixed shuffle
    // tree -> deterministic for a given nblk
    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (only_sum && t != 11) return;                       // the pAp pass: cost partials of the previous update stay where they are
    double s = 0.;
    for (int b = lane; b < nblk; b += 64) s += partials[(size_t)b * 12 + t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) out[t] = s;
}

bool labeldot_streaming(const tnml_ctx* c, int NTp) {
    // few images on this rank: 64-image workgroups with 16 (fp64) waves each, so that the chip still has enough loads in flight
    const int force = c->opt_ldot_cfg;                        // 1: streaming form, 2: small-shard form (env TNML_LDOT_CFG / tnml_set_option "ldot_cfg")
    return force ? force == 1 : NTp / 128 >= 192;
}
int launch_labeldot_blocks(tnml_ctx* c, const LdotArgs& a_in, int blk_off, int nblk, hipStream_t st, int kclass, int form) {
    // the Label-carrying operand is read exactly once per launch: non-temporal loads stream it at 6.4-6.5 TB/s
    // instead of 5.6-5.7 with the default cache policy (profiles/r01_ab_nt_loads.txt)
    LdotArgs a = a_in;
    a.nt = 1; a.blk_off = blk_off;
    const bool small = form ? form == 2 : !labeldot_streaming(c, a.NTp);
    if (blk_off + nblk > c->partial_cap) return tnml_fail(c, "labeldot: partial buffer too small");
    if (!st) st = c->stream;
    ProfScope ps(c, kclass, st);             // the streaming kernel alone: bench.py's HBM roofline divides by these launches
#define LDOT(NW, IPL, TA, TB, TC) do { if (a.nl == 1) hipLaunchKernelGGL((k_labeldot<NW, IPL, 1, TA, TB, TC>), dim3(nblk), dim3(64 * NW), 0, st, a, c->partials); \
                                        else hipLaunchKernelGGL((k_labeldot<NW, IPL, TNML_NL, TA, TB, TC>), dim3(nblk), dim3(64 * NW), 0, st, a, c->partials); } while (0)
    if (c->env64() && form == 3) {          // 128-image blocks, two rows in flight per wave (split launches)
        if (a.nl == 1) hipLaunchKernelGGL((k_labeldot<4, 2, 1, double, double, double, 2>), dim3(nblk), dim3(256), 0, st, a, c->partials);
        else           hipLaunchKernelGGL((k_labeldot<4, 2, TNML_NL, double, double, double, 2>), dim3(nblk), dim3(256), 0, st, a, c->partials);
    } else if (c->env64()) {
        if (small) LDOT(16, 1, double, double, double); else LDOT(4, 2, double, double, double);
    } else if (c->f64()) {
        if (a.a_is_env) { if (small) LDOT(16, 1, float, double, double); else LDOT(4, 2, float, double, double); }
        else            { if (small) LDOT(16, 1, double, float, double); else LDOT(4, 2, double, float, double); }
    } else {
        if (small) LDOT(16, 1, float, float, float); else LDOT(8, 2, float, float, float);
    }
#undef LDOT
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_labeldot_reduce(tnml_ctx* c, int nblk_total, double* scal_out, int only_sum) {
    ProfScope ps(c, KC_PUPDATE);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(768), 0, c->stream, c->partials, nblk_total, scal_out, only_sum);
    HIPCK(c, hipGetLastError());
    return 0;
}
int launch_labeldot(tnml_ctx* c, const LdotArgs& a, double* scal_out, bool reduce) {
    const int nblk = a.NTp / (labeldot_streaming(c, a.NTp) ? 128 : 64);
    TCK(launch_labeldot_blocks(c, a, 0, nblk, c->stream, KC_LABELDOT, 0));
    c->part_n = nblk;
    if (!reduce) return 0;
    return launch_labeldot_reduce(c, nblk, scal_out, a.mode == LD_MODE_PAP ? 1 : 0);
}

template <typename T, typename TE>
__global__ void k_zprime_t(const TE* __restrict__ EL, size_t lstride, const T* __restrict__ dP,
                           T* __restrict__ Z, int mq, int NTp) {
    const size_t n2 = (size_t)NTp / 2;
    const size_t total = (size_t)mq * n2;
    typedef typename vec2<T>::type T2;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t q = idx / n2, n = (idx % n2) * 2;
        T zx = 0, zy = 0;
#pragma unroll
        for (int l = 0; l < TNML_NL; ++l) {
            const typename vec2<TE>::type e = *reinterpret_cast<const typename vec2<TE>::type*>(EL + (size_t)l * lstride + q * NTp + n);
            const T2 d = *reinterpret_cast<const T2*>(dP + (size_t)l * NTp + n);
            zx = fma((T)e.x, d.x, zx); zy = fma((T)e.y, d.y, zy);
        }
        T2 z; z.x = zx; z.y = zy;
        *reinterpret_cast<T2*>(Z + q * NTp + n) = z;
    }
}

int launch_zprime(tnml_ctx* c, const void* EL, size_t lstride, const void* dP, void* Z, int mq, int NTp) {
    ProfScope ps(c, KC_ZPRIME);
    const size_t total = (size_t)mq * (NTp / 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (c->env64())    hipLaunchKernelGGL((k_zprime_t<double, double>), dim3(blocks), dim3(256), 0, c->stream, (const double*)EL, lstride, (const double*)dP, (double*)Z, mq, NTp);
    else if (c->f64()) hipLaunchKernelGGL((k_zprime_t<double, float>), dim3(blocks), dim3(256), 0, c->stream, (const float*)EL, lstride, (const double*)dP, (double*)Z, mq, NTp);
    else               hipLaunchKernelGGL((k_zprime_t<float, float>), dim3(blocks), dim3(256), 0, c->stream, (const float*)EL, lstride, (const float*)dP, (float*)Z, mq, NTp);
    HIPCK(c, hipGetLastError());
    return 0;
}

// Fast CG (tnml_ctx::fast_cg): B*t.v is linear in B, so after B <- B + a p the model outputs are
// P <- P + a (p*t.v) with p*t.v already computed by the pAp pass (the idea of the reference's own
// single.h:290-398 fast_cgrad).  Recomputes dP, the per-label cost partials and the argmax count.
template <typename T>
__global__ __launch_bounds__(LD_IMGS) void k_pupdate(T* __restrict__ P, const T* __restrict__ Pp, T* __restrict__ dP,
                                                    const int* __restrict__ label, int NTp,
                                                    const double* __restrict__ alpha, const double* __restrict__ conv,
                                                    double* __restrict__ partials, int nl, int target) {
    if (conv[0] != 0.) return;                             // CG already converged: P must stay as it is
    __shared__ double s_part[(LD_IMGS / 64) * 12];
    pupdate_unit<T>(P, Pp, dP, label, NTp, (T)alpha[0], partials, nl, target, s_part, threadIdx.x, blockIdx.x);
}

int launch_pupdate(tnml_ctx* c, const double* alpha_dev, double* scal_out, bool reduce) {
    ProfScope ps(c, KC_PUPDATE);
    const int nblk = c->NTp / LD_IMGS;
    c->part_n = nblk;
    if (c->f64()) hipLaunchKernelGGL(k_pupdate<double>, dim3(nblk), dim3(LD_IMGS), 0, c->stream, (double*)c->P, (const double*)c->Pp, (double*)c->dP, c->label, c->NTp, alpha_dev, c->scal + SC_CONVP + ((c->cg_pass - 1) & 1), c->partials, c->nl(), c->target());
    else          hipLaunchKernelGGL(k_pupdate<float>, dim3(nblk), dim3(LD_IMGS), 0, c->stream, (float*)c->P, (const float*)c->Pp, (float*)c->dP, c->label, c->NTp, alpha_dev, c->scal + SC_CONVP + ((c->cg_pass - 1) & 1), c->partials, c->nl(), c->target());
    if (reduce) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(768), 0, c->stream, c->partials, nblk, scal_out, 0);
    HIPCK(c, hipGetLastError());
    return 0;
}

template <typename TE>
__global__ void k_features_u8(const uint8_t* __restrict__ pix, int N, int NT, int NTp, TE* __restrict__ phi) {
    // pix [NT][N] -> phi [N][2][NTp]; one thread per (site, image)
    const size_t total = (size_t)N * NTp;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx / NTp), n = (int)(idx % NTp);
        TE f0 = 0, f1 = 0;
        if (n < NT) {
            // g = byte/255. (mllib/mnist.h:495); x = g/255.; phi = pow(x/4., n-1) (fixedL.cc:640-641),
            // evaluated in fp64 then rounded once to fp32
            const double g = (double)pix[(size_t)n * N + j] / 255.;
            f0 = 1;
            f1 = (TE)((g / 255.) / 4.);
        }
        phi[((size_t)j * 2 + 0) * NTp + n] = f0;
        phi[((size_t)j * 2 + 1) * NTp + n] = f1;
    }
}

int launch_features_u8(tnml_ctx* c, const uint8_t* d_pix, int N, int NT, int NTp, void* phi) {
    ProfScope ps(c, KC_PACK);
    if (c->env64()) hipLaunchKernelGGL(k_features_u8<double>, dim3(4096), dim3(256), 0, c->stream, d_pix, N, NT, NTp, (double*)phi);
    else            hipLaunchKernelGGL(k_features_u8<float>, dim3(4096), dim3(256), 0, c->stream, d_pix, N, NT, NTp, (float*)phi);
    HIPCK(c, hipGetLastError());
    return 0;
}
