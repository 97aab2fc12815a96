// fulltest_main.cpp -- the `fulltest <inputfile>` evaluator on top of the C-ABI (include/tnml.h).
//
// Keeps the reference's surface (fulltest.cc:7-100): keys `datadir`, `fname` (default "W"), `feature`
// (series | normal), the files `sites` and <fname> in the working directory, the t10k idx files under
// `datadir`, and the result table of fullTest (util.h:186-199).  The per-image contraction toverlap
// (util.h:19-40) is one tnml_classify call on the device.  Extensions: `device`, `precision`
// (f64 | mixed | f32), `Ntest` (per-label cap; the reference takes the whole test set), `imglen` and
// `feature_scale` as in the fixedL driver (they must match the values W was trained with).
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

static void die(tnml_ctx* c, const char* what) {
    std::fprintf(stderr, "%s: %s\n", what, tnml_last_error(c));
    std::exit(1);
}
#define CK(c, call) do { if ((call) != 0) die((c), #call); } while (0)

int main(int argc, const char* argv[]) {
    if (argc != 2) { std::printf("Usage: %s inputfile\n", argv[0]); return 0; }       // fulltest.cc:10-14
    try {
        InputGroup input(argv[1], "input");
        const int d = 2;
        const std::string datadir = input.getString("datadir", "/Users/mstoudenmire/software/tnml/mllib/MNIST");
        const std::string fname = input.getString("fname", "W");
        const std::string feature = input.getString("feature", "series");
        const int device = (int)input.getInt("device", 0);
        const std::string precision = input.getString("precision", "f64");
        const long Ntest = input.getInt("Ntest", 50000);                               // mllib/mnist.h:452 default NT
        const long imglen = input.getInt("imglen", 0);
        const double feature_scale = input.getReal("feature_scale", 1.);
        int dtype = TNML_F64;
        if (precision == "mixed") dtype = TNML_F64_E32; else if (precision == "f32") dtype = TNML_F32;
        else if (precision != "f64" && precision != "strict") { std::printf("precision must be f64, mixed or f32\n"); return 1; }

        std::printf("Labels:"); for (int l = 0; l < 10; ++l) std::printf(" %d", l); std::printf("\n");   // :28
        Dataset test = read_mnist(datadir, false, Ntest);                              // :30
        if (imglen > 0) reduce(test, (int)imglen);
        const int N = test.npix();
        if (!file_exists("sites")) { std::printf("Couldn't find file 'sites'\n"); return 1; }             // :34-41
        int Ns, ds; read_sites("sites", &Ns, &ds);
        if (Ns != N || ds != d) { std::printf("Mismatched sizes\n"); return 1; }                         // util.h:68
        bool normal;
        if (feature == "norm" || feature == "normal") normal = true;                   // :45-56
        else if (feature == "series") normal = false;
        else { std::printf("feature type \"%s\" not recognized\n", feature.c_str()); return 1; }

        std::printf("Converting test set to MPS\n");                                   // :74
        const int totNtest = test.size();
        std::printf("Total of %d testing images\n", totNtest);                         // :85
        if (!file_exists(fname)) { std::printf("Couldn't find file '%s'\n", fname.c_str()); return 1; }   // :88-95
        HostMPS psi = read_mps(fname);
        if (psi.N != N) { std::printf("Mismatched sizes\n"); return 1; }
        int cent = 0;                                                                  // util.h:128-139
        for (int j = 1; j <= N; ++j) if (psi.A[j].L == NL) { cent = j; break; }
        if (cent == 0) { std::printf("expected Label index at some site of psi MPS\n"); return 1; }
        if (cent != N / 2) { std::printf("Label Index not on site %d\n", N / 2); return 1; }
        int wm = 1; for (int j = 1; j <= N; ++j) wm = std::max(wm, std::max(psi.A[j].ml, psi.A[j].mr));

        tnml_config cfg{};
        cfg.device = device; cfg.rank = 0; cfg.nranks = 1; cfg.N = N; cfg.NT_local = totNtest; cfg.NT_total = totNtest;
        cfg.maxm = wm; cfg.dtype = dtype; cfg.svd_backend = TNML_SVD_SYEVD;
        tnml_ctx* ctx = nullptr;
        if (tnml_create(&ctx, &cfg)) die(nullptr, "tnml_create");
        if (!normal && !test.reduced() && feature_scale == 1.) {
            CK(ctx, tnml_set_data_u8(ctx, test.pixels.data(), test.labels.data()));    // phi = [1, x/4], x = (byte/255)/255
        } else {
            std::vector<double> phi = all_features(test, normal, feature_scale);       // fulltest.cc:57-70
            CK(ctx, tnml_set_data_phi(ctx, phi.data(), test.labels.data()));
        }
        for (int j = 1; j <= N; ++j) CK(ctx, tnml_set_site(ctx, j, psi.A[j].ml, psi.A[j].mr, psi.A[j].L == NL, psi.A[j].a.data()));

        std::printf("Running full test of %s\n", fname.c_str());                       // :97
        int64_t counts[10], ninc[10];
        CK(ctx, tnml_classify(ctx, nullptr, nullptr, counts, ninc));
        long nte = 0, tninc = 0;
        for (int l = 0; l < 10; ++l) { nte += (long)counts[l]; tninc += (long)ninc[l]; }
        const long tncor = nte - tninc;
        std::printf("%ld/%ld correct (%.2f%%), %ld/%ld incorrect (%.2f%%)\n",           // util.h:186-187
                    tncor, nte, tncor * 100. / nte, tninc, nte, tninc * 100. / nte);
        long tot = 0;
        for (int l = 0; l < 10; ++l) {                                                 // util.h:189-198
            const long nt = (long)counts[l]; tot += nt;
            if (nt == 0) continue;
            const long ni = (long)ninc[l], nc = nt - ni;
            std::printf("  Digit %d %ld/%ld correct (%.2f%%), %ld/%ld incorrect (%.2f%%)\n", l, nc, nt, nc * 100. / nt, ni, nt, ni * 100. / nt);
        }
        std::printf("Total # test images = %ld\n", tot);                               // util.h:199
        tnml_destroy(ctx);
    } catch (const std::exception& e) {
        std::printf("%s\n", e.what());
        return 1;
    }
    return 0;
}
