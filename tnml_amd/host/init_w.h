// init_w.h -- initial weight MPS of fixedL.cc:702-728: per label, the truncated sum of `ninitial`
// random training product states of that label (Cutoff 1E-10, Maxm 10), Label index attached on site
// N/2 with weight 0.1, then the truncated sum over the ten labels (Cutoff 1E-8, Maxm 10), centre tensor
// normalised.  The reference draws images with ITensor's time-seeded Global::random() (util.h:104-121);
// here the seed is explicit (input key `seed`, an extension) so runs are reproducible.
#pragma once
#include <cmath>
#include <random>

#include "host_mps.h"
#include "mnist_idx.h"

namespace tnmlh {

// phi(g,n) = pow((g/255.)/4., n-1) with g = byte/255. (mllib/mnist.h:495, fixedL.cc:637-642); `scale` multiplies the
// second component (1 = the reference's double normalisation, 255 = the README's [1, x/4] -- SURVEY.md 9-Q1)
inline void features_series(const Dataset& d, int img, std::vector<double>& phi, double scale = 1.) {
    const int N = d.npix();
    phi.resize((size_t)N * 2);
    for (int j = 0; j < N; ++j) { const double g = d.value(img, j) / 255.; phi[2 * j] = 1.; phi[2 * j + 1] = scale * ((g / 255.) / 4.); }
}
// fulltest.cc:57-66 "normal" map with the same double normalisation: x = g/255
inline void features_normal(const Dataset& d, int img, std::vector<double>& phi) {
    const int N = d.npix();
    phi.resize((size_t)N * 2);
    for (int j = 0; j < N; ++j) { const double x = (d.value(img, j) / 255.) / 255.; phi[2 * j] = std::cos(M_PI / 2. * x); phi[2 * j + 1] = std::sin(M_PI / 2. * x); }
}
// all images: [n][N][2], the layout of tnml_set_data_phi
inline std::vector<double> all_features(const Dataset& d, bool normal, double scale) {
    const int N = d.npix();
    std::vector<double> out((size_t)d.size() * N * 2), phi;
    for (int i = 0; i < d.size(); ++i) {
        if (normal) features_normal(d, i, phi); else features_series(d, i, phi, scale);
        std::copy(phi.begin(), phi.end(), out.begin() + (size_t)i * N * 2);
    }
    return out;
}

// util.h:104-121 randImg: uniform index, retry (<= 1000 times) until the label matches
inline int rand_img(const Dataset& d, int label, std::mt19937_64& rng) {
    std::uniform_real_distribution<double> u(0., 1.);
    for (int t = 0; t < 1000; ++t) {
        long w = (long)(d.size() * u(rng));
        if (w < 0) w = 0;
        if (w >= d.size()) w = d.size() - 1;
        if (d.labels[w] == label) return (int)w;
    }
    throw std::runtime_error("Did not find image with requested label after 1000 tries");
}

inline HostMPS sum_truncated(const std::vector<HostMPS>& v, double cutoff, int maxm) {
    HostMPS acc = v.at(0);
    for (size_t k = 1; k < v.size(); ++k) { acc = add(acc, v[k]); compress(acc, cutoff, maxm); }
    if (v.size() == 1) compress(acc, cutoff, maxm);
    return acc;
}

// single.cc:112-128: W = sum of `ninitial` random product states of label `label` (Cutoff 1E-10, Maxm 10), orthogonalised,
// normalised, orthogonality centre on site 1 (W.position(1,...) is then a no-op)
inline HostMPS build_initial_single(const Dataset& train, int label, int ninitial, uint64_t seed, bool normal, double feature_scale = 1.) {
    const int N = train.npix();
    std::mt19937_64 rng(seed);
    std::vector<HostMPS> psis;
    std::vector<double> phi;
    for (int m = 0; m < ninitial; ++m) {
        const int w = rand_img(train, label, rng);
        if (normal) features_normal(train, w, phi); else features_series(train, w, phi, feature_scale);
        psis.push_back(product_state(N, phi.data()));                     // :119
    }
    HostMPS W = psis.at(0);
    for (size_t k = 1; k < psis.size(); ++k) { W = add(W, psis[k]); compress(W, 1E-10, 10, true); }   // :122 sum(psis,{"Cutoff",1E-10,"Maxm",10})
    if (psis.size() == 1) compress(W, 1E-10, 10, true);                   // :123 orthogonalize
    const double nrm = norm_site(W.A[1]);                                 // :124 (centre on site 1: norm(A_1) = norm(W))
    for (double& x : W.A[1].a) x /= nrm;
    return W;
}

inline HostMPS build_initial_w(const Dataset& train, int ninitial, uint64_t seed, bool verbose, double feature_scale = 1.) {
    const int N = train.npix();
    std::mt19937_64 rng(seed);
    std::vector<HostMPS> ipsis;
    std::vector<double> phi;
    for (int n = 0; n < NL; ++n) {
        std::vector<HostMPS> psis;
        for (int m = 0; m < ninitial; ++m) {
            const int w = rand_img(train, n, rng);
            features_series(train, w, phi, feature_scale);
            psis.push_back(product_state(N, phi.data()));                 // :717
        }
        if (verbose) printf("Summing %d random label %d states\n", ninitial, n);   // :719
        HostMPS s = sum_truncated(psis, 1E-10, 10);                       // :720
        attach_label(s, n, 0.1);                                          // :721
        ipsis.push_back(std::move(s));
    }
    if (verbose) printf("Summing all %d label states together\n", (int)ipsis.size());   // :723
    HostMPS W = sum_truncated(ipsis, 1E-8, 10);                           // :724
    const double nrm = norm_site(W.A[W.c0]);                              // :725
    for (double& x : W.A[W.c0].a) x /= nrm;
    return W;
}

}  // namespace tnmlh
