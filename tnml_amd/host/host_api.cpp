// host_api.cpp -- C entry points over the host-side driver pieces (input parser, idx reader, weight
// files, initial-W builder) so that they can be unit-tested from Python without a GPU
// (tests/test_host_driver.py).  Built into tnml_amd/libtnml_host.so; the fixedL binary uses the same
// headers directly.
#include <cstring>
#include <string>

#include "host_mps.h"
#include "init_w.h"
#include "input_group.h"
#include "mnist_idx.h"

using namespace tnmlh;
static thread_local std::string g_err;
extern "C" {
const char* tnmlh_last_error() { return g_err.c_str(); }
// value of `key` in group `input` of `file` as a string ("" + return 1 if absent)
int tnmlh_input_get(const char* file, const char* key, char* out, int cap) {
    try { InputGroup g(file, "input"); if (!g.has(key)) { out[0] = 0; return 1; } std::strncpy(out, g.getString(key, "").c_str(), cap - 1); out[cap - 1] = 0; return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int tnmlh_input_yesno(const char* file, const char* key, int def) {
    try { InputGroup g(file, "input"); return g.getYesNo(key, def != 0) ? 1 : 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// idx reader: sizes first (pixels == NULL), then the data
int tnmlh_read_mnist(const char* datadir, int train, long nt_per_label, int* n, int* npix, unsigned char* pixels, int* labels, long* file_index) {
    try {
        Dataset d = read_mnist(datadir, train != 0, nt_per_label);
        *n = d.size(); *npix = d.npix();
        if (pixels) std::memcpy(pixels, d.pixels.data(), d.pixels.size());
        if (labels) std::memcpy(labels, d.labels.data(), sizeof(int) * d.labels.size());
        if (file_index) std::memcpy(file_index, d.file_index.data(), sizeof(long) * d.file_index.size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// block-mean down-sampling (mnist_idx.h reduce) of n square images of side `side` to newlen x newlen; out[n][newlen^2]
int tnmlh_reduce(const unsigned char* pixels, int n, int side, int newlen, double* out) {
    try {
        Dataset d; d.rows = d.cols = side; d.pixels.assign(pixels, pixels + (size_t)n * side * side); d.labels.assign(n, 0);
        reduce(d, newlen);
        for (size_t k = 0; k < (size_t)n * newlen * newlen; ++k) out[k] = d.value(0, k);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// initial W from a dataset directory -> file `out` (TNMLW1); returns overlap(W,W) through *ovl
int tnmlh_build_initial_w(const char* datadir, long nt_per_label, int ninitial, unsigned long long seed, const char* out, double* ovl, int* maxdim,
                          int imglen, double feature_scale) {
    try {
        Dataset d = read_mnist(datadir, true, nt_per_label);
        if (imglen > 0) reduce(d, imglen);
        HostMPS W = build_initial_w(d, ninitial, seed, false, feature_scale);
        write_mps(out, W);
        if (ovl) *ovl = overlap(W, W);
        int md = 1; for (int j = 1; j <= W.N; ++j) md = std::max(md, std::max(W.A[j].ml, W.A[j].mr));
        if (maxdim) *maxdim = md;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// initial W of the per-label variant (single.cc:112-128) -> file `out`
int tnmlh_build_initial_single(const char* datadir, long nt_per_label, int label, int ninitial, unsigned long long seed, int normal,
                               const char* out, int imglen, double feature_scale) {
    try {
        Dataset d = read_mnist(datadir, true, nt_per_label);
        if (imglen > 0) reduce(d, imglen);
        write_mps(out, build_initial_single(d, label, ninitial, seed, normal != 0, feature_scale));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// `sites` file (SiteSet(N,d), fixedL.cc:619-631)
int tnmlh_sites_write(const char* file, int N, int d) { try { write_sites(file, N, d); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; } }
int tnmlh_sites_read(const char* file, int* N, int* d) { try { read_sites(file, N, d); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; } }
// weight file access: dims of site j, then its data (column-major [ml][2][mr][L])
int tnmlh_mps_info(const char* file, int* N, int* c0) {
    try { HostMPS W = read_mps(file); *N = W.N; *c0 = W.c0; return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int tnmlh_mps_site(const char* file, int j, int* ml, int* mr, int* L, double* data) {
    try { HostMPS W = read_mps(file); const Site& s = W.A.at(j); *ml = s.ml; *mr = s.mr; *L = s.L; if (data) std::memcpy(data, s.a.data(), sizeof(double) * s.a.size()); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// write a weight file from a flat list of sites (dims[3*j..] = ml, mr, L; data concatenated)
int tnmlh_mps_write(const char* file, int N, const int* dims, const double* data) {
    try {
        HostMPS W(N); size_t off = 0;
        for (int j = 1; j <= N; ++j) { W.A[j] = Site(dims[3 * (j - 1)], dims[3 * (j - 1) + 1], dims[3 * (j - 1) + 2]); std::memcpy(W.A[j].a.data(), data + off, sizeof(double) * W.A[j].a.size()); off += W.A[j].a.size(); }
        write_mps(file, W); return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
}
