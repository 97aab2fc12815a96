// separate_fulltest_main.cpp -- the `separate_fulltest <inputfile>` evaluator of the per-label variant on top of the
// C-ABI (TNML_MODE_SINGLE, tnml_classify).
//
// Keeps the reference's surface (separate_fulltest.cc:7-170): keys `datadir`, `fname`, `imglen` (read, unused there),
// the file `sites`, the ten weight files `L<n>/W<n>`, the t10k idx files, the "normal" feature map (hard-coded in the
// reference, :104-118), and the printed tables: per image the ten overlaps o_n = <W_n|x>, prediction
// argmax_n |o_n| (first maximum), costs[n] += (n == l) ? (o_n - 1)^2 : o_n^2.  Extensions: `feature` (normal | series),
// `device`, `precision`, `Ntest`, `imglen` honoured as block-mean down-sampling, `feature_scale`.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/tnml.h"
#include "host_mps.h"
#include "init_w.h"
#include "input_group.h"
#include "mnist_idx.h"

using namespace tnmlh;

static void die(tnml_ctx* c, const char* what) { std::fprintf(stderr, "%s: %s\n", what, tnml_last_error(c)); std::exit(1); }
#define CK(c, call) do { if ((call) != 0) die((c), #call); } while (0)

int main(int argc, const char* argv[]) {
    if (argc != 2) { std::printf("Usage: %s inputfile\n", argv[0]); return 0; }       // :89-93
    try {
        InputGroup input(argv[1], "input");
        const std::string datadir = input.getString("datadir", "/Users/mstoudenmire/software/tnml/mllib/MNIST");
        (void)input.getString("fname", "W");
        const long imglen = input.getInt("imglen", 28);                                // :99
        const std::string feature = input.getString("feature", "normal");
        const int device = (int)input.getInt("device", 0);
        const std::string precision = input.getString("precision", "f64");
        const long Ntest = input.getInt("Ntest", 50000);
        const double feature_scale = input.getReal("feature_scale", 1.);
        int dtype = TNML_F64;
        if (precision == "mixed") dtype = TNML_F64_E32; else if (precision == "f32") dtype = TNML_F32;
        else if (precision != "f64" && precision != "strict") { std::printf("precision must be f64, mixed or f32\n"); return 1; }
        bool normal;
        if (feature == "normal") normal = true; else if (feature == "series") normal = false;
        else { std::printf("feature=%s not recognized\n", feature.c_str()); return 1; }
        const int NLW = 10;
        std::printf("Labels:"); for (int l = 0; l < NLW; ++l) std::printf(" %d", l); std::printf("\n");   // :108

        Dataset test = read_mnist(datadir, false, Ntest);                              // :124
        if (imglen > 0 && imglen < test.rows) reduce(test, (int)imglen);
        const int N = test.npix();
        if (!file_exists("sites")) { std::printf("Couldn't find file 'sites'\n"); return 1; }             // :128-135
        int Ns, ds; read_sites("sites", &Ns, &ds);
        if (Ns != N || ds != 2) { std::printf("Mismatched sizes\n"); return 1; }
        std::printf("Converting test set to MPS\n");                                   // :137
        const int totNtest = test.size();
        std::printf("Total of %d testing images\n", totNtest);                         // :150

        std::vector<HostMPS> Ws(NLW);
        int wm = 1;
        for (int n = 0; n < NLW; ++n) {                                                // :156-160
            char path[64]; std::snprintf(path, sizeof path, "L%d/W%d", n, n);
            Ws[n] = read_mps(path);
            if (Ws[n].N != N) { std::printf("Mismatched sizes\n"); return 1; }
            for (int j = 1; j <= N; ++j) { if (Ws[n].A[j].L != 1) { std::printf("%s carries a Label index\n", path); return 1; }
                                           wm = std::max(wm, std::max(Ws[n].A[j].ml, Ws[n].A[j].mr)); }
        }
        tnml_config cfg{};
        cfg.device = device; cfg.rank = 0; cfg.nranks = 1; cfg.N = N; cfg.NT_local = totNtest; cfg.NT_total = totNtest;
        cfg.maxm = wm; cfg.dtype = dtype; cfg.svd_backend = TNML_SVD_SYEVD; cfg.mode = TNML_MODE_SINGLE; cfg.target_label = 0;
        tnml_ctx* ctx = nullptr;
        if (tnml_create(&ctx, &cfg)) die(nullptr, "tnml_create");
        {
            std::vector<double> phi = all_features(test, normal, feature_scale);
            CK(ctx, tnml_set_data_phi(ctx, phi.data(), test.labels.data()));
        }
        std::printf("Running full test\n");                                            // :165
        std::vector<std::vector<double>> o(NLW, std::vector<double>(totNtest));        // o[n][image] = overlap(Ws[n], testimg), :38
        for (int n = 0; n < NLW; ++n) {
            for (int j = 1; j <= N; ++j) CK(ctx, tnml_set_site(ctx, j, Ws[n].A[j].ml, Ws[n].A[j].mr, 0, Ws[n].A[j].a.data()));
            CK(ctx, tnml_classify(ctx, o[n].data(), nullptr, nullptr, nullptr));
        }
        long counts[10] = {0}, ninc[10] = {0}, tninc = 0, tncor = 0, ntest = 0;
        double costs[10] = {0};
        for (int i = 0; i < totNtest; ++i) {                                           // fullTest, :9-59 (order independent sums)
            const int l = test.labels[i];
            counts[l]++; ++ntest;
            int pl = 0; double best = std::fabs(o[0][i]);
            for (int n = 0; n < NLW; ++n) {
                const double on = o[n][i];
                costs[n] += (n == l) ? (on - 1) * (on - 1) : on * on;                  // :40
                if (n > 0 && std::fabs(on) > best) { best = std::fabs(on); pl = n; }   // :39,44 first maximum
            }
            if (pl == l) ++tncor; else { ++tninc; ++ninc[l]; }
        }
        std::printf("%ld/%ld correct (%.2f%%), %ld/%ld incorrect (%.2f%%)\n", tncor, ntest, tncor * 100. / ntest, tninc, ntest, tninc * 100. / ntest);   // :61
        long tot = 0;
        for (int l = 0; l < 10; ++l) {                                                 // :64-73
            const long nt = counts[l]; tot += nt;
            if (nt == 0) continue;
            const long ni = ninc[l], nc = nt - ni;
            std::printf("  Digit %d %ld/%ld correct (%.2f%%), %ld/%ld incorrect (%.2f%%)\n", l, nc, nt, nc * 100. / nt, ni, nt, ni * 100. / nt);
        }
        std::printf("Total # test images = %ld\n", tot);                               // :74
        double tC = 0.;
        std::printf("Cost functions:\n");                                              // :77
        for (int l = 0; l < 10; ++l) { tC += costs[l]; std::printf("  Digit %d C = %.20f\n", l, costs[l]); }   // :81
        std::printf("Total C = %.20f\n", tC);                                          // :83
        tnml_destroy(ctx);
    } catch (const std::exception& e) {
        std::printf("%s\n", e.what());
        return 1;
    }
    return 0;
}
