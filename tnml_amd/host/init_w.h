// init_w.h -- initial weight MPS of fixedL.cc:702-728: per label, the truncated sum of `ninitial`
// random training product states of that label (Cutoff 1E-10, Maxm 10), Label index attached on site
// N/2 with weight 0.1, then the truncated sum over the ten labels (Cutoff 1E-8, Maxm 10), centre tensor
// normalised.  The reference draws images with ITensor's time-seeded Global::random() (util.h:104-121);
// here the seed is explicit (input key `seed`, an extension) so runs are reproducible.
#pragma once
#include <random>

#include "host_mps.h"
#include "mnist_idx.h"

namespace tnmlh {

// phi(g,n) = pow((g/255.)/4., n-1) with g = byte/255. (mllib/mnist.h:495, fixedL.cc:637-642)
inline void features_series(const uint8_t* pix, int N, std::vector<double>& phi) {
    phi.resize((size_t)N * 2);
    for (int j = 0; j < N; ++j) { const double g = pix[j] / 255.; phi[2 * j] = 1.; phi[2 * j + 1] = (g / 255.) / 4.; }
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

inline HostMPS build_initial_w(const Dataset& train, int ninitial, uint64_t seed, bool verbose) {
    const int N = train.npix();
    std::mt19937_64 rng(seed);
    std::vector<HostMPS> ipsis;
    std::vector<double> phi;
    for (int n = 0; n < NL; ++n) {
        std::vector<HostMPS> psis;
        for (int m = 0; m < ninitial; ++m) {
            const int w = rand_img(train, n, rng);
            features_series(&train.pixels[(size_t)w * N], N, phi);
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
