// single_main.cpp -- the `single <inputfile>` command line driver of the per-label variant on top of the C-ABI
// (include/tnml.h, TNML_MODE_SINGLE).
//
// Keeps the reference's surface (single.cc:6-244 and the mldmrg of single.h:523-728): input keys, the files `sites`
// and `W<label>` in the working directory, the `WRITE_WF` hook, the idx-ubyte training set under `datadir`, the
// image order (labels taken round-robin, single.cc:156-181) and the log lines.  One tnml_bond_update per bond.
// Built: method = conj | fast_conj | exact | pinv and the `noise` density-matrix split (single.h:648-672).  method = pinv is what it is in the
// reference (single.h:596-604): the cost of the pseudo-inverse solution is printed, the update itself is cgrad; its random start comes from
// `seed` here (the reference's is time-seeded).  Extensions (never read by the reference): `seed`, `device`, `precision`, `imglen`,
// `feature_scale` as in the fixedL driver; `labels`, `ngpu`, `share_device`, `dry_run`: the one-label-per-GPU launcher
// of BASELINE config 4 (launch_per_label below).
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <array>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <iostream>
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

static HostMPS download(tnml_ctx* ctx, int N) {
    HostMPS W(N);
    for (int j = 1; j <= N; ++j) {
        int ml, mr, hl; CK(ctx, tnml_site_dims(ctx, j, &ml, &mr, &hl));
        W.A[j] = Site(ml, mr, 1);
        CK(ctx, tnml_get_site(ctx, j, W.A[j].a.data()));
    }
    return W;
}

// BASELINE config 4: "single.cc per-label MPS x10, one label per GPU (embarrassingly parallel, no collectives)".  The reference trains
// the ten networks as ten runs of `single` with ten input files, each in its own directory L<n> (separate_fulltest.cc:158 reads
// L%d/W%d).  Extension key `labels = all` (or a comma list): this process becomes a launcher -- one child `single` per label, the
// label and the device handed over in the environment, working directory L<label>, at most one running child per GPU (key `ngpu`,
// default: every visible device), the next label taken as soon as a device is free; `share_device = yes` keeps all children on
// `device` one after the other (one-GPU boxes); `dry_run = yes` prints the plan only.  Replicas only: no collective anywhere.
static int launch_per_label(const char* self, const char* inputfile, const InputGroup& input) {
    std::vector<int> labels;
    const std::string spec = input.getString("labels", "");
    if (spec == "all") for (int l = 0; l < 10; ++l) labels.push_back(l);
    else {
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t e = spec.find(',', pos); if (e == std::string::npos) e = spec.size();
            const int l = std::atoi(spec.substr(pos, e - pos).c_str());
            if (l < 0 || l > 9) { std::printf("labels: %d is not in 0..9\n", l); return 1; }
            labels.push_back(l); pos = e + 1;
        }
    }
    if (labels.empty()) { std::printf("labels: nothing to do\n"); return 1; }
    const bool dry = input.getYesNo("dry_run", false), share = input.getYesNo("share_device", false);
    const int dev0 = (int)input.getInt("device", 0);
    int ngpu = (int)input.getInt("ngpu", 0);
    if (share) ngpu = 1;
    if (ngpu <= 0) {
        if (dry) ngpu = 8;
        else { int64_t f, t; ngpu = 0; while (tnml_device_memory(dev0 + ngpu, &f, &t) == 0) ++ngpu; }
        if (ngpu == 0) { std::fprintf(stderr, "no HIP device: %s\n", tnml_last_error(nullptr)); return 1; }
    }
    char inabs[PATH_MAX], selfabs[PATH_MAX], ddabs[PATH_MAX];
    if (!realpath(inputfile, inabs)) { std::perror(inputfile); return 1; }
    if (!realpath(self, selfabs)) std::snprintf(selfabs, sizeof selfabs, "%s", self);
    const std::string dd = input.getString("datadir", "");
    const bool have_dd = !dd.empty() && realpath(dd.c_str(), ddabs) != nullptr;
    std::printf("Per-label training of %zu labels on %d GPU%s (one `single` process per label, no collectives)\n", labels.size(), ngpu, ngpu == 1 ? "" : "s");
    std::vector<pid_t> busy(ngpu, 0); std::vector<int> busy_label(ngpu, -1);
    int failed = 0;
    size_t next = 0, running = 0;
    auto reap = [&](bool block) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, block ? 0 : WNOHANG);
        if (pid <= 0) return false;
        for (int d = 0; d < ngpu; ++d) if (busy[d] == pid) {
            const bool ok = WIFEXITED(st) && WEXITSTATUS(st) == 0;
            std::printf("label %d on device %d: %s (log: L%d/log)\n", busy_label[d], dev0 + d, ok ? "done" : "FAILED", busy_label[d]);
            if (!ok) ++failed;
            busy[d] = 0; busy_label[d] = -1; --running;
        }
        return true;
    };
    while (next < labels.size() || running) {
        int d = -1;
        if (dry) d = (int)(next % (size_t)ngpu);                    // the plan: as if every training took the same time
        else for (int k = 0; k < ngpu && next < labels.size(); ++k) if (!busy[k]) { d = k; break; }
        if (d < 0) { reap(true); continue; }
        const int l = labels[next++];
        char dir[16]; std::snprintf(dir, sizeof dir, "L%d", l);
        std::printf("label %d -> device %d, directory %s, writes %s/W%d\n", l, dev0 + d, dir, dir, l);
        if (dry) continue;
        std::fflush(stdout);
        mkdir(dir, 0777);
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) {
            if (chdir(dir) != 0) _exit(111);
            if (!freopen("log", "w", stdout)) _exit(112);
            char buf[32];
            std::snprintf(buf, sizeof buf, "%d", l); setenv("TNML_SINGLE_LABEL", buf, 1);
            std::snprintf(buf, sizeof buf, "%d", dev0 + d); setenv("TNML_SINGLE_DEVICE", buf, 1);
            if (have_dd) setenv("TNML_SINGLE_DATADIR", ddabs, 1);
            execl(selfabs, selfabs, inabs, (char*)nullptr);
            _exit(113);
        }
        busy[d] = pid; busy_label[d] = l; ++running;
    }
    if (dry) return 0;
    std::printf("%zu of %zu per-label trainings finished%s\n", labels.size() - failed, labels.size(), failed ? "" : "; evaluate with `separate_fulltest <inputfile>` from this directory");
    return failed ? 1 : 0;
}

int main(int argc, const char* argv[]) {
    if (argc != 2) { std::printf("Usage: %s inputfile\n", argv[0]); return 0; }       // single.cc:11-15
    try {
        InputGroup input(argv[1], "input");
        const char* env_label = std::getenv("TNML_SINGLE_LABEL");                       // set by the launcher for its children
        if (!env_label && !input.getString("labels", "").empty()) return launch_per_label(argv[0], argv[1], input);
        const char* env_dd = std::getenv("TNML_SINGLE_DATADIR");
        const std::string datadir = env_dd ? std::string(env_dd) : input.getString("datadir", "/Users/mstoudenmire/software/tnml/mllib/MNIST");
        const int L = env_label ? std::atoi(env_label) : (int)input.getInt("label", 0);
        const long Ntrain = input.getInt("Ntrain", 60000);
        const long Nsweep = input.getInt("Nsweep", 50);
        const double cutoff = input.getReal("cutoff", 1E-8);
        const long maxm = input.getInt("maxm", 5000);
        const long minm = input.getInt("minm", std::max(10L, maxm / 2));
        const double noise = input.getReal("noise", 0.);
        const long ninitial = input.getInt("ninitial", 100);
        const long Nthread = input.getInt("nthread", 4);
        const bool pause_steps = input.getYesNo("pause_steps", false);
        const std::string feature = input.getString("feature", "normal");
        bool normal;
        if (feature == "normal") normal = true; else if (feature == "series") normal = false;
        else { std::printf("feature=%s not recognized\n", feature.c_str()); return 1; }   // :33-38
        const double lambda = input.getReal("lambda", 0.);
        const std::string method = input.getString("method", "conj");
        (void)input.getReal("alpha", 1.0); (void)input.getReal("clip", 1.0);
        const long Npass = input.getInt("Npass", 4);
        const double cconv = input.getReal("cconv", 1E-10);
        const long Ntarget = input.getInt("Ntarget", 10); const double pcut = input.getReal("pcut", 1E-8); (void)input.getYesNo("precalc", true);
        const uint64_t seed = (uint64_t)input.getInt("seed", 1);
        const int device = std::getenv("TNML_SINGLE_DEVICE") ? std::atoi(std::getenv("TNML_SINGLE_DEVICE")) : (int)input.getInt("device", 0);
        const std::string precision = input.getString("precision", "f64");
        const long imglen = input.getInt("imglen", 0);
        const double feature_scale = input.getReal("feature_scale", 1.);
        int dtype = TNML_F64;
        if (precision == "mixed") dtype = TNML_F64_E32; else if (precision == "f32") dtype = TNML_F32;
        else if (precision != "f64" && precision != "strict") { std::printf("precision must be f64, mixed or f32\n"); return 1; }
        if (L < 0 || L > 9) { std::printf("label must be in 0..9\n"); return 1; }
        if (method != "conj" && method != "fast_conj" && method != "exact" && method != "pinv") { std::printf("method type \"%s\" not recognized\n", method.c_str()); return 1; }   // single.h:611
        const bool fast_conj = method == "fast_conj";                                  // single.h:599
        const bool exact = method == "exact";                                          // single.h:600
        const bool pinv = method == "pinv";                                            // single.h:596

        char wname[32]; std::snprintf(wname, sizeof wname, "W%d", L);                  // :53
        Dataset train = read_mnist(datadir, true, Ntrain);                              // :56
        if (imglen > 0) reduce(train, (int)imglen);
        const int N = train.npix();
        std::printf("%d sites\n", N);                                                  // :59
        if (file_exists("sites")) { int Ns, ds; read_sites("sites", &Ns, &ds); if (Ns != N || ds != 2) { std::printf("Mismatched sizes\n"); return 1; } }
        else write_sites("sites", N, 2);                                                // :61-69
        std::printf("Converting training set to MPS\n");                                // :87
        const int totNtrain = train.size();
        const int totL = train.counts[L];
        std::printf("Total of %d training images\n", totNtrain);                        // :101
        std::printf("%d training images with selected label L=%d\n", totL, L);          // :102

        HostMPS W;
        if (file_exists(wname)) {                                                       // :106-110
            std::printf("Reading %s from file\n", wname);
            W = read_mps(wname);
            if (W.N != N) { std::printf("Mismatched sizes\n"); return 1; }
            for (int j = 1; j <= N; ++j) if (W.A[j].L != 1) { std::printf("%s carries a Label index\n", wname); return 1; }
        } else {
            std::printf("Summing %ld random label %d states\n", ninitial, L);           // :121
            W = build_initial_single(train, L, (int)ninitial, seed, normal, feature_scale);
        }
        std::printf("Done making initial W\n");                                         // :127
        std::printf("Thread %d %d -> %d (%d)\n", 0, 0, totNtrain, totNtrain);           // :147-150 (one GPU in place of Nthread threads)
        (void)Nthread;

        // ts order: labels round-robin, each label's images in file order (single.cc:156-181)
        std::vector<std::vector<int>> by_label(10);
        for (int i = 0; i < totNtrain; ++i) by_label[train.labels[i]].push_back(i);
        std::vector<int> order; order.reserve(totNtrain);
        {
            std::array<size_t, 10> ncount{};
            int nl = 0;
            for (int k = 0; k < totNtrain; ++k) {
                int count = 0, pick = -1;
                while (pick < 0) {
                    const int l = nl; nl = (nl + 1 == 10) ? 0 : nl + 1;
                    if (ncount[l] < by_label[l].size()) pick = by_label[l][ncount[l]++];
                    if (++count > 20 && pick < 0) { std::printf("Infinite loop while setting up ts\n"); return 1; }
                }
                order.push_back(pick);
            }
        }
        std::vector<int32_t> labels(totNtrain);
        std::vector<double> phi((size_t)totNtrain * N * 2), ph;
        for (int k = 0; k < totNtrain; ++k) {
            labels[k] = train.labels[order[k]];
            if (normal) features_normal(train, order[k], ph); else features_series(train, order[k], ph, feature_scale);
            std::copy(ph.begin(), ph.end(), phi.begin() + (size_t)k * N * 2);
        }
        int wm = 1; for (int j = 1; j <= N; ++j) wm = std::max(wm, std::max(W.A[j].ml, W.A[j].mr));

        tnml_config cfg{};
        cfg.device = device; cfg.rank = 0; cfg.nranks = 1; cfg.N = N; cfg.NT_local = totNtrain; cfg.NT_total = totNtrain;
        cfg.maxm = (int)std::min<long>(maxm, 1 << 20); cfg.dtype = dtype; cfg.svd_backend = TNML_SVD_SYEVD;
        cfg.mode = TNML_MODE_SINGLE; cfg.target_label = L;
        {   // `maxm` is only an upper bound for the reference: size the context by what N sites can reach and the GPU can hold
            int64_t freeb = 0, totb = 0;
            if (tnml_device_memory(device, &freeb, &totb) != 0) die(nullptr, "tnml_device_memory");
            cfg.maxm = std::max(wm, tnml_plan_maxm(&cfg, cfg.maxm, wm, (int64_t)(0.97 * (double)freeb)));
            if (cfg.maxm < maxm) std::printf("maxm=%ld is beyond what %d sites can reach or the GPU can hold for %d images: bond dimensions are capped at %d\n", maxm, N, totNtrain, cfg.maxm);
        }
        tnml_ctx* ctx = nullptr;
        if (tnml_create(&ctx, &cfg) != 0) die(nullptr, "tnml_create");
        if (fast_conj) CK(ctx, tnml_set_option(ctx, "cg_method", 1));
        if (exact) { CK(ctx, tnml_set_option(ctx, "cg_method", 2)); CK(ctx, tnml_set_option_real(ctx, "pcut", pcut)); }
        if (noise >= 1E-14) CK(ctx, tnml_set_option_real(ctx, "noise", noise));           // sweeps.noise() = noise, single.cc:222
        CK(ctx, tnml_set_data_phi(ctx, phi.data(), labels.data()));
        phi.clear(); phi.shrink_to_fit();
        for (int j = 1; j <= N; ++j) CK(ctx, tnml_set_site(ctx, j, W.A[j].ml, W.A[j].mr, 0, W.A[j].a.data()));
        std::printf("Projecting training states..."); std::fflush(stdout);              // :183
        CK(ctx, tnml_env_init(ctx));                                                    // :184-199
        std::printf("done\n");
        {
            int mL, mR, lab; CK(ctx, tnml_bond_dims(ctx, 1, &mL, &mR, &lab));
            std::vector<double> B((size_t)mL * 4 * mR);
            CK(ctx, tnml_bond_tensor(ctx, 1, B.data()));
            double C;
            CK(ctx, tnml_quadcost(ctx, B.data(), lambda, &C, nullptr, nullptr, nullptr));   // :217
            std::printf("Before DMRG, Cost = %.10f\n", C / Ntrain);                     // :218 (divides by the per-label cap, as the reference)
        }
        const double NT = (double)totNtrain;                                            // single.h:535 Ntrain = ts.size()
        for (long sw = 1; sw <= Nsweep; ++sw) {                                         // single.h:546
            std::printf("Sweep %ld maxm=%ld\n", sw, maxm);                              // :548
            for (int b = 1, ha = 1; ha <= 2; tnml_sweepnext(&b, &ha, N)) {              // :554
                tnml_sweep_params sp{(int)std::min<long>(maxm, cfg.maxm), (int)std::min<long>(minm, cfg.maxm), cutoff, (int)Npass, lambda, lambda, cconv, 1};
                tnml_bond_report r;
                double pinv_cost = 0.; std::vector<double> pve, pD; int pdone = 0;
                if (pinv) {                                                             // single.h:596-601: BB = B; pinv(BB,...); quadcost(BB)
                    int mL, mR, lab; CK(ctx, tnml_set_bond(ctx, b)); CK(ctx, tnml_bond_dims(ctx, b, &mL, &mR, &lab));
                    const int D = mL * 4 * mR, rr = (int)std::min<long>(std::min<long>(Ntarget, D), 64);
                    std::vector<double> V0((size_t)D * rr), BB((size_t)D);
                    uint64_t st = (uint64_t)seed * 0x9E3779B97F4A7C15ull + (uint64_t)(sw * 100003 + ha * 1009 + b);   // random(...) of :457, seeded
                    for (double& x : V0) { st = st * 6364136223846793005ull + 1442695040888963407ull; x = (double)((st >> 11) & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5; }
                    pve.assign((size_t)Npass + 1, 0.); pD.assign((size_t)rr, 0.);
                    CK(ctx, tnml_pinv(ctx, V0.data(), rr, (int)Npass, lambda, pcut, BB.data(), pve.data(), &pdone, pD.data()));
                    CK(ctx, tnml_quadcost(ctx, BB.data(), lambda, &pinv_cost, nullptr, nullptr, nullptr));
                }
                CK(ctx, tnml_bond_update(ctx, b, ha, &sp, &r));
                std::printf("Sweep %ld Half %d Bond %d\n", sw, ha, r.c);                // :566
                std::printf("norm(oB) = %.12g\n", r.norm_oB);                           // :572
                if (pinv) {
                    std::printf("Using pcut = %.2E\n", pcut);                             // :415
                    std::printf("Initial V*E = %.20f\n", pve[0]);                         // :476
                    for (int p = 1; p <= pdone; ++p) { std::printf("Making E\nPolar U\n%d V*E = %.20f\n", p, pve[(size_t)p]); }   // :481,489,498
                    std::printf("D ="); for (double x : pD) std::printf(" %.10g", x); std::printf("\n");   // :508 PrintData(D)
                    std::printf("After pinv, Cost = %.20f\n", pinv_cost / NT);             // :601
                }
                if (r.cg.converged == 2) std::printf("  |r| < %.1E, not optimizing\n", cconv);   // :204 (|r| itself stays on the device)
                for (int p = 0; fast_conj && p < r.cg.npass_done; ++p) {                // fast_cgrad prints pass and |r| on one line, no cost (single.h:337,373,387-393)
                    std::printf("  Conj grad pass %d ", p + 1);
                    const bool has_r = r.cg.converged ? true : p + 1 < r.cg.npass_done || r.cg.npass_done < Npass;
                    if (!(has_r && p + 1 < Npass)) std::printf("\n");
                    else if (r.cg.converged == 1 && p + 1 == r.cg.npass_done) std::printf("  |r| = %.1E < %.1E, breaking\n", r.cg.rnorm[p], cconv);
                    else std::printf("  |r| = %.1E\n", r.cg.rnorm[p]);
                }
                for (int p = 0; !fast_conj && p < r.cg.npass_done; ++p) {
                    std::printf("  Conj grad pass %d\n", p + 1);                        // :211
                    const bool has_cost = r.cg.converged ? true : p + 1 < r.cg.npass_done || r.cg.npass_done < Npass;
                    if (has_cost && (p + 1 < Npass)) {
                        std::printf("  %d C = %.10f\n", p + 1, r.cg.cost[p] / NT);      // :271
                        if (r.cg.converged == 1 && p + 1 == r.cg.npass_done) std::printf("  |r| = %.1E < %.1E, breaking\n", r.cg.rnorm[p], cconv);   // :275
                        else std::printf("  |r| = %.1E\n", r.cg.rnorm[p]);              // :280
                    }
                }
                std::printf("Sweep %ld Half %d Bond %d\n", sw, ha, r.c);                // :618
                std::printf("Cost = %.10f --> %.10f\n", r.cost_old / NT, r.cost_cg / NT);   // :623
                if (lambda > 0.) {
                    std::printf("Reg. cost RC = %.10f (%.10f)\n", r.reg_cost_cg / NT, r.reg_cost_cg);                       // :627
                    std::printf("Cost - RC = %.10f (%.10f)\n", (r.cost_cg - r.reg_cost_cg) / NT, r.cost_cg - r.reg_cost_cg);   // :628
                }
                if (noise < 1E-14) std::printf("SVD trunc err = %.2E\n", r.truncerr);   // :646
                else               std::printf("Trunc err = %.2E\n", r.truncerr);       // :669 (density-matrix split)
                std::printf("Original m=%d, New m=%d\n", r.origm, r.newm);              // :678
                std::printf("norm(newB) = %.12g\n", r.norm_newB);                       // :681
                std::printf("--> After SVD, Cost = %.10f (%.10f)\n", r.cost_after_svd / NT, r.cost_after_svd);   // :684
                if (r.cost_after_svd > 1.1 * r.cost_cg) std::printf("> 10%% larger C after SVD\n");   // :686
                if (pause_steps) { std::printf("PAUSE"); std::fflush(stdout); std::getchar(); }
                if (file_exists("WRITE_WF")) {                                          // :712-718
                    std::printf("File WRITE_WF found\n");
                    std::remove("WRITE_WF");
                    std::printf("Writing %s to disk\n", wname);
                    write_mps(wname, download(ctx, N));
                }
                std::fflush(stdout);
            }
            std::printf("Writing %s to disk\n", wname);                                 // :722
            write_mps(wname, download(ctx, N));
        }
        std::printf("Writing %s to disk\n", wname);                                     // single.cc:240
        write_mps(wname, download(ctx, N));
        tnml_destroy(ctx);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
