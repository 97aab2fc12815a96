// fixedl_main.cpp -- the `fixedL <inputfile>` command line driver on top of the C-ABI (include/tnml.h).
//
// Keeps the reference's surface (fixedL.cc:573-767): the input-file grammar and keys, the files in the
// working directory (sites, W, WRITE_WF, LAMBDA), the idx-ubyte dataset under `datadir`, and the log
// lines, while the sweep itself (mldmrg, fixedL.cc:451-570) is one
// tnml_bond_update call per bond.  Extensions (never read by the reference, all optional): `seed`
// (initial-W RNG), `device` (HIP ordinal), `precision` (f64 = fp64 everywhere [default], mixed = fp64 MFMA over
// fp32-stored environments, f32 = fp32 MFMA study mode), `imglen` (block-mean down-sampling of the images to
// imglen x imglen; present in the reference's sample input but never read by fixedL.cc), `feature_scale`
// (multiplies the second feature component; 1 = the reference's double normalisation).
#include <array>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

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

static void upload(tnml_ctx* ctx, const HostMPS& W) {
    for (int j = 1; j <= W.N; ++j) CK(ctx, tnml_set_site(ctx, j, W.A[j].ml, W.A[j].mr, W.A[j].L == NL, W.A[j].a.data()));
}
static HostMPS download(tnml_ctx* ctx, int N) {
    HostMPS W(N);
    for (int j = 1; j <= N; ++j) {
        int ml, mr, hl; CK(ctx, tnml_site_dims(ctx, j, &ml, &mr, &hl));
        W.A[j] = Site(ml, mr, hl ? NL : 1);
        CK(ctx, tnml_get_site(ctx, j, W.A[j].a.data()));
    }
    return W;
}

int main(int argc, const char* argv[]) {
    if (argc != 2) { std::printf("Usage: %s inputfile\n", argv[0]); return 0; }       // fixedL.cc:579-583
    try {
        InputGroup input(argv[1], "input");                                             // :584
        const int d = 2;
        const std::string datadir = input.getString("datadir", "/Users/mstoudenmire/software/tnml/mllib/MNIST");
        const long Ntrain = input.getInt("Ntrain", 60000);
        const long Nbatch = input.getInt("Nbatch", 10);
        const long Nsweep = input.getInt("Nsweep", 50);
        const double cutoff = input.getReal("cutoff", 1E-10);
        const long maxm = input.getInt("maxm", 5000);
        const long minm = input.getInt("minm", std::max(10L, maxm / 2));
        const long ninitial = input.getInt("ninitial", 100);
        const long Nthread = input.getInt("nthread", 1);
        (void)input.getYesNo("replace", false);                                         // read, never used
        const bool pause_step = input.getYesNo("pause_step", false);
        double lambda = input.getReal("lambda", 0.);
        const std::string method = input.getString("method", "conj");
        (void)input.getReal("alpha", 0.01); (void)input.getReal("clip", 1.0);           // passed, never read
        const long Npass = input.getInt("Npass", 4);
        const double cconv = input.getReal("cconv", 1E-10);
        const uint64_t seed = (uint64_t)input.getInt("seed", 1);                         // extension
        const int device = (int)input.getInt("device", 0);                              // extension
        const std::string precision = input.getString("precision", "f64");              // extension: f64 | mixed | f32
        const long imglen = input.getInt("imglen", 0);                                   // extension: 0 = keep the file's size
        const double feature_scale = input.getReal("feature_scale", 1.);                 // extension
        int dtype = TNML_F64;
        if (precision == "mixed") dtype = TNML_F64_E32; else if (precision == "f32") dtype = TNML_F32;
        else if (precision != "f64" && precision != "strict") { std::printf("precision must be f64, mixed or f32\n"); return 1; }
        if (method != "conj") { std::printf("method type \"%s\" not recognized\n", method.c_str()); return 1; }   // :505

        Dataset train = read_mnist(datadir, true, Ntrain);                              // :613
        if (imglen > 0) reduce(train, (int)imglen);
        std::printf("Training set consists of %d images:\n", train.size());
        for (int l = 0; l < 10; ++l) std::printf("  %d of label %d\n", train.counts[l], l);
        const int N = train.npix();                                                     // :615
        const int c = N / 2;                                                            // :616
        std::printf("%d sites of dimension %d\n", N, d);                                // :617
        if (file_exists("sites")) {                                                     // :619-627
            int Ns, ds; read_sites("sites", &Ns, &ds);
            if (ds != d) { std::printf("Error: d=%d but dimension of first site is %d\n", d, ds); return 1; }
            if (Ns != N) { std::printf("Error: sites file has %d sites, data has %d\n", Ns, N); return 1; }
        } else write_sites("sites", N, d);                                              // :630-631
        std::printf("Converting training set to MPS\n");                                // :644
        const int totNtrain = train.size();
        std::printf("Total of %d training images\n", totNtrain);                        // :655
        if (totNtrain % Nbatch != 0) {                                                  // :84-89
            std::printf("totNtrain=%d, Nbatch=%ld, totNtrain%%Nbatch=%ld\n", totNtrain, Nbatch, totNtrain % Nbatch);
            std::printf("totNtrain not commensurate with Nbatch\n");
            return 1;
        }
        std::printf("Thread %d %d -> %d (%d)\n", 0, 0, totNtrain, totNtrain);           // :94 (one GPU in place of Nthread threads)
        (void)Nthread;

        HostMPS W;
        if (file_exists("W")) {                                                         // :671-681
            std::printf("Reading W from disk\n");
            W = read_mps("W");
            if (W.N != N || W.A[c].L != NL) { std::printf("Expected W to have Label type Index at site %d\n", c); return 1; }
        } else if (file_exists("W0")) {                                                  // :682-701
            std::printf("Found separate W0,W1,...,W9 MPS: summing\n");
            std::vector<HostMPS> ipsis;
            for (int n = 0; n < 10; ++n) {
                char fn[16]; std::snprintf(fn, sizeof fn, "W%d", n);
                HostMPS in = read_mps(fn);                                              // :691 per-label weight MPS (single / linear)
                if (in.N != N) { std::printf("%s has %d sites, data has %d\n", fn, in.N, N); return 1; }
                for (int j = 1; j <= N; ++j) if (in.A[j].L != 1) { std::printf("%s already carries a Label index\n", fn); return 1; }
                in.c0 = c;
                attach_label(in, n, 1.0);                                               // :693 in.Aref(c) *= setElt(Lval(n))
                ipsis.push_back(std::move(in));
            }
            std::printf("Summing all %d label states together\n", (int)ipsis.size());  // :696
            W = sum_truncated(ipsis, 1E-10, 1 << 30);                                   // :697 sum(ipsis,{"Cutoff",1E-10})
            std::printf("Done making initial W\n");
            write_mps("W", W);                                                          // :700
        } else {
            W = build_initial_w(train, (int)ninitial, seed, true, feature_scale);                      // :702-726
            std::printf("Done making initial W\n");
            write_mps("W", W);                                                          // :727
        }
        std::printf("overlap(W,W) = %.12g\n", overlap(W, W));                           // :729
        for (int j = 1; j <= N; ++j) if ((W.A[j].L == NL) != (j == c)) { std::printf("Label Index not on site %d\n", c); return 1; }   // :734
        int wm = 1; for (int j = 1; j <= N; ++j) wm = std::max(wm, std::max(W.A[j].ml, W.A[j].mr));

        tnml_config cfg{};
        cfg.device = device; cfg.rank = 0; cfg.nranks = 1; cfg.N = N; cfg.NT_local = totNtrain; cfg.NT_total = totNtrain;
        cfg.maxm = (int)std::max<long>(std::min<long>(maxm, 4096), wm); cfg.dtype = dtype; cfg.svd_backend = TNML_SVD_SYEVD;
        tnml_ctx* ctx = nullptr;
        if (tnml_create(&ctx, &cfg) != 0) die(nullptr, "tnml_create");
        if (!train.reduced() && feature_scale == 1.) {
            CK(ctx, tnml_set_data_u8(ctx, train.pixels.data(), train.labels.data()));   // TState ctor, :644-653
        } else {
            std::vector<double> phi = all_features(train, false, feature_scale);
            CK(ctx, tnml_set_data_phi(ctx, phi.data(), train.labels.data()));
        }
        upload(ctx, W);
        std::printf("Projecting training states..."); std::fflush(stdout);              // :740
        CK(ctx, tnml_env_init(ctx));                                                    // :741
        std::printf("done\n");
        std::printf("Calling quadcost...\n");                                           // :744
        {
            int mL, mR, lab; CK(ctx, tnml_bond_dims(ctx, 1, &mL, &mR, &lab));
            std::vector<double> B((size_t)mL * 4 * mR * (lab ? NL : 1));
            CK(ctx, tnml_bond_tensor(ctx, 1, B.data()));
            double C, lc[10], cr; int64_t nc;
            CK(ctx, tnml_quadcost(ctx, B.data(), lambda, &C, lc, &cr, &nc));            // :745
            std::printf("Percent correct = %.4f%%, # incorrect = %lld/%d\n", nc * 100. / totNtrain, (long long)(totNtrain - nc), totNtrain);
            std::printf("Before starting DMRG Cost = %.10f\n", C / totNtrain);          // :746
        }
        if (pause_step) { std::printf("PAUSE"); std::fflush(stdout); std::getchar(); }

        const double lambda_cost = lambda;                                              // cargs copy, :467
        for (long sw = 1; sw <= Nsweep; ++sw) {                                         // mldmrg, :470
            std::printf("\nSweep %ld maxm=%ld minm=%ld\n", sw, maxm, minm);            // :472
            for (int b = 1, ha = 1; ha <= 2; tnml_sweepnext(&b, &ha, N)) {              // :478
                tnml_sweep_params sp{(int)std::min<long>(maxm, cfg.maxm), (int)minm, cutoff, (int)Npass, lambda, lambda_cost, cconv, 0};
                tnml_bond_report r;
                CK(ctx, tnml_bond_update(ctx, b, ha, &sp, &r));
                std::printf("Sweep %ld Half %d Bond %d\n", sw, ha, r.c);                // :490
                std::printf("In cgrad, lambda = %.3E\n", lambda);                       // :358
                for (int p = 0; p < r.cg.npass_done; ++p) {
                    std::printf("  Conj grad pass %d\n", p + 1);                        // :391
                    const bool has_cost = r.cg.converged ? true : p + 1 < r.cg.npass_done || r.cg.npass_done < Npass;
                    if (has_cost && (p + 1 < Npass)) {
                        std::printf("  Cost = %.10f\n", r.cg.cost[p] / totNtrain);      // :429
                        if (r.cg.converged && p + 1 == r.cg.npass_done) std::printf("  |r| = %.1E < %.1E, breaking\n", r.cg.rnorm[p], cconv);   // :434
                        else std::printf("  |r| = %.1E\n", r.cg.rnorm[p]);              // :439
                    }
                }
                std::printf("Sweep %ld Half %d Bond %d\n", sw, ha, r.c);                // :510
                std::printf("SVD trunc err = %.2E\n", r.truncerr);                      // :523
                std::printf("Original m=%d, New m=%d\n", r.origm, r.newm);              // :525
                std::printf("norm(newB) = %.12g\n", r.norm_newB);                       // :528
                std::printf("rank(newB) = %d\n", r.label_on_B ? 5 : 4);                 // :529 (tensor order)
                std::printf("|B-newB| = %.3E\n", r.diff_B_newB);                        // :530
                for (int l = 0; l < 10; ++l) std::printf("  Label l=%d C%d = %.10f\n", l, l, r.label_cost[l] / totNtrain);   // :334
                std::printf("  Reg. cost CR = %.10f\n", r.reg_cost / totNtrain);        // :337
                std::printf("Percent correct = %.4f%%, # incorrect = %lld/%d\n", r.ncorrect * 100. / totNtrain,
                            (long long)(totNtrain - r.ncorrect), totNtrain);            // :341-342
                std::printf("--> After SVD, Cost = %.10f\n", r.cost_after_svd / totNtrain);   // :533
                const int cs = ha == 1 ? b : b + 1, prevc = ha == 1 ? b - 1 : b + 2;    // :196-209
                if (prevc >= 1 && prevc <= N) std::printf("## Advancing E from %d to %d\n", prevc, cs);
                else std::printf("## Making new E at %d\n", cs);
                if (file_exists("WRITE_WF")) {                                          // :542-548
                    std::printf("File WRITE_WF found\n");
                    std::remove("WRITE_WF");
                    std::printf("Writing W to disk\n");
                    write_mps("W", download(ctx, N));
                }
                if (file_exists("LAMBDA")) {                                            // :550-559
                    std::ifstream lf("LAMBDA"); lf >> lambda; lf.close();
                    std::remove("LAMBDA");
                    std::cout << "new lambda = " << lambda << std::endl;
                }
                if (pause_step) { std::printf("PAUSE"); std::fflush(stdout); std::getchar(); }   // :561
                std::fflush(stdout);
            }
            std::printf("Writing W to disk\n");                                         // :565
            write_mps("W", download(ctx, N));                                           // :566
        }
        std::printf("Writing W to disk\n");                                             // :763
        write_mps("W", download(ctx, N));                                               // :764
        tnml_destroy(ctx);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
