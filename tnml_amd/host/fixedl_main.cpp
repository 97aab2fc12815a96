// fixedl_main.cpp -- the `fixedL <inputfile>` command line driver on top of the C-ABI (include/tnml.h).
//
// Keeps the reference's surface (fixedL.cc:573-767): the input-file grammar and keys, the files in the
// working directory (sites, W, WRITE_WF, LAMBDA), the idx-ubyte dataset under `datadir`, and the log
// lines, while the sweep itself (mldmrg, fixedL.cc:451-570) is one
// tnml_bond_update call per bond.  Multi-GPU: `ngpu = n` shards the training images over n GPUs (ParallelDo's chunk rule,
// paralleldo.h:32-43), one host thread and one context per GPU inside this process, gradient / cost sums over RCCL; rank 0
// prints and writes files, so the log of an n-GPU run is the log of the 1-GPU run.  Extensions (never read by the
// reference, all optional): `seed` (initial-W RNG), `device` (first HIP ordinal), `ngpu`, `precision` (f64 = fp64 everywhere [default], mixed = fp64 MFMA over
// fp32-stored environments, f32 = fp32 MFMA study mode), `imglen` (block-mean down-sampling of the images to
// imglen x imglen; present in the reference's sample input but never read by fixedL.cc), `feature_scale`
// (multiplies the second feature component; 1 = the reference's double normalisation), `pipeline`, `bond_log`
// (a CSV line per bond update: cost, #correct, bond dimensions, truncation error, seconds).
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
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

// the ranks of one process meet here (a generation-counting barrier)
class HostBarrier {
    std::mutex mu; std::condition_variable cv; int n, waiting = 0; long gen = 0;
  public:
    explicit HostBarrier(int n_) : n(n_) {}
    void wait() {
        if (n <= 1) return;
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

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
        const int device = (int)input.getInt("device", 0);                              // extension: first HIP ordinal
        const long ngpu = input.getInt("ngpu", 1);                                      // extension: GPUs (ranks) to shard the images over; 0 = all visible
        const bool share_device = input.getYesNo("share_device", false);                // extension: all ranks on `device` (in-process communicator; one-GPU boxes)
        const std::string allreduce_kind = input.getString("allreduce", "rccl");        // extension: "oneshot" = peer-write all-gather + rank-ordered local sum (tnml_comm_init_oneshot)
        const bool oneshot = allreduce_kind == "oneshot";
        if (!oneshot && allreduce_kind != "rccl") die(nullptr, "allreduce must be rccl or oneshot");
        const std::string bond_log = input.getString("bond_log", "");                      // extension: a CSV line per bond update (machine-readable log for parity and bond updates/s)
        const double env_budget_gb = input.getReal("env_budget_gb", 0.);                    // extension: cap on the environments held in HBM, the rest lives in host memory (the reference's Nbatch / proj_images spill); 0 = all resident
        const bool pipeline = input.getYesNo("pipeline", true);                         // extension: enqueue bond k+1 before fetching the report of bond k (see the sweep loop)
        const std::string precision = input.getString("precision", "f64");              // extension: f64 | mixed | f32 | bf16x3 | bf16
        const long imglen = input.getInt("imglen", 0);                                   // extension: 0 = keep the file's size
        const double feature_scale = input.getReal("feature_scale", 1.);                 // extension
        int dtype = TNML_F64;
        if (precision == "mixed") dtype = TNML_F64_E32; else if (precision == "f32") dtype = TNML_F32;
        else if (precision == "bf16x3") dtype = TNML_BF16X3; else if (precision == "bf16") dtype = TNML_BF16;   // study modes (forward contraction on the bf16 matrix pipe)
        else if (precision != "f64" && precision != "strict") { std::printf("precision must be f64, mixed, f32, bf16x3 or bf16\n"); return 1; }
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

        // ---- ranks: one host thread per GPU (the reference's `nthread` worker threads become GPUs: paralleldo.h:21-68) ----
        int nranks = (int)ngpu;
        if (nranks <= 0 && share_device) nranks = 1;
        if (nranks <= 0) {                                                              // ngpu = 0: every visible device from `device` on
            nranks = 0;
            int64_t f, t;
            while (tnml_device_memory(device + nranks, &f, &t) == 0) ++nranks;
            if (nranks == 0) die(nullptr, "tnml_device_memory");
        }
        if (nranks > totNtrain) nranks = totNtrain;
        std::vector<int64_t> lo(nranks), hi(nranks);
        for (int r = 0; r < nranks; ++r) {
            tnml_shard_bounds(totNtrain, nranks, r, &lo[r], &hi[r]);
            std::printf("Thread %d %lld -> %lld (%lld)\n", r, (long long)lo[r], (long long)hi[r], (long long)(hi[r] - lo[r]));   // :94, one GPU per "thread"
        }
        (void)Nthread;
        // `maxm` is only an upper bound for the reference (default 5000): the contexts are sized by what an N-site MPS can
        // reach and what every GPU can hold for its shard
        int ctx_maxm = (int)std::min<long>(maxm, 1 << 20);
        for (int r = 0; r < nranks; ++r) {
            tnml_config pc{}; pc.device = share_device ? device : device + r; pc.rank = r; pc.nranks = nranks; pc.N = N; pc.NT_local = (int)(hi[r] - lo[r]); pc.NT_total = totNtrain;
            pc.maxm = ctx_maxm; pc.dtype = dtype;
            int64_t freeb = 0, totb = 0;
            if (tnml_device_memory(pc.device, &freeb, &totb) != 0) die(nullptr, "tnml_device_memory");
            if (share_device) freeb /= nranks;
            // (with a host tier the environments need not fit: only what an N-site MPS can reach bounds maxm then)
            ctx_maxm = std::min(ctx_maxm, tnml_plan_maxm(&pc, ctx_maxm, wm, env_budget_gb > 0. ? (int64_t)0 : (int64_t)(0.97 * (double)freeb)));
        }
        ctx_maxm = std::max(ctx_maxm, wm);
        if (ctx_maxm < maxm)
            std::printf("maxm=%ld is beyond what %d sites can reach or the GPU can hold for %lld images: bond dimensions are capped at %d\n",
                        maxm, N, (long long)(hi[0] - lo[0]), ctx_maxm);

        std::vector<double> phi_all;
        const bool use_u8 = !train.reduced() && feature_scale == 1.;
        if (!use_u8) phi_all = all_features(train, false, feature_scale);
        unsigned char uid[128] = {0};
        if (nranks > 1 && !share_device && !oneshot && tnml_comm_unique_id(uid) != 0) die(nullptr, "tnml_comm_unique_id");
        std::vector<tnml_ctx*> all_ctx(nranks, nullptr);

        HostBarrier bar(nranks);
        double lambda_shared = lambda;
        bool write_wf = false;
        auto rank_main = [&](int r) {
            const bool root = r == 0;
            tnml_config cfg{};
            cfg.device = share_device ? device : device + r; cfg.rank = r; cfg.nranks = nranks; cfg.N = N; cfg.NT_local = (int)(hi[r] - lo[r]); cfg.NT_total = totNtrain;
            cfg.maxm = ctx_maxm; cfg.dtype = dtype; cfg.svd_backend = TNML_SVD_SYEVD;
            tnml_ctx* ctx = nullptr;
            if (tnml_create(&ctx, &cfg) != 0) die(nullptr, "tnml_create");
            if (env_budget_gb > 0.) CK(ctx, tnml_set_option(ctx, "env_budget_mb", (int)(env_budget_gb * 1024.)));
            if (use_u8) CK(ctx, tnml_set_data_u8(ctx, train.pixels.data() + (size_t)lo[r] * N, train.labels.data() + lo[r]));   // TState ctor, :644-653
            else        CK(ctx, tnml_set_data_phi(ctx, phi_all.data() + (size_t)lo[r] * N * 2, train.labels.data() + lo[r]));
            if (nranks > 1 && !share_device && !oneshot) CK(ctx, tnml_comm_init(ctx, uid));    // RCCL over xGMI, one rank per GPU
            if (nranks > 1 && (share_device || oneshot)) {                                  // in-process communicators: staging buffer on one GPU, or one-shot peer writes
                all_ctx[r] = ctx;
                bar.wait();
                if (root && (oneshot ? tnml_comm_init_oneshot(all_ctx.data(), nranks) : tnml_comm_init_local(all_ctx.data(), nranks)) != 0) die(nullptr, "in-process communicator");
                bar.wait();
            }
            upload(ctx, W);
            if (nranks > 1) { int cnt = 0; CK(ctx, tnml_replica_check(ctx, &cnt)); if (root) std::printf("%s communicator of %d ranks, W replicas identical\n", oneshot ? "one-shot peer-write" : (share_device ? "in-process" : "RCCL"), cnt); }
            if (root) { std::printf("Projecting training states..."); std::fflush(stdout); }   // :740
            CK(ctx, tnml_env_init(ctx));                                                    // :741
            if (root) { std::printf("done\n"); std::printf("Calling quadcost...\n"); }     // :744
            {
                int mL, mR, lab; CK(ctx, tnml_bond_dims(ctx, 1, &mL, &mR, &lab));
                std::vector<double> B((size_t)mL * 4 * mR * (lab ? NL : 1));
                CK(ctx, tnml_bond_tensor(ctx, 1, B.data()));
                double C, lc[10], cr; int64_t nc;
                CK(ctx, tnml_quadcost(ctx, B.data(), lambda, &C, lc, &cr, &nc));            // :745
                if (root) {
                    std::printf("Percent correct = %.4f%%, # incorrect = %lld/%d\n", nc * 100. / totNtrain, (long long)(totNtrain - nc), totNtrain);
                    std::printf("Before starting DMRG Cost = %.10f\n", C / totNtrain);      // :746
                }
            }
            if (root && pause_step) { std::printf("PAUSE"); std::fflush(stdout); std::getchar(); }
            bar.wait();

            double lam = lambda;
            const double lambda_cost = lambda;                                              // cargs copy, :467
            // One bond update is enqueued (tnml_bond_update_begin) before the report of the previous one is fetched
            // (tnml_bond_update_end), so the GPU never waits for the host at a bond boundary and, with several ranks, the previous
            // bond's cost partials ride in this bond's first all-reduce.  The log lines of a bond therefore appear while the next one
            // runs, and the WRITE_WF / LAMBDA hooks act one bond later than in the reference (input key `pipeline = no`, or
            // pause_step, restores the strict order).  Nothing is in flight across a sweep boundary.
            struct InFlight { long sw; int b, ha; double lam; bool on = false; } fl;
            FILE* blog = (root && !bond_log.empty()) ? std::fopen(bond_log.c_str(), "w") : nullptr;
            if (blog) std::fprintf(blog, "sweep,half,bond,lambda,cg_passes,cost_after_svd,reg_cost,ncorrect,ntrain,orig_m,new_m,trunc_err,seconds\n");
            auto t_last = std::chrono::steady_clock::now();
            auto finish = [&]() {                                                           // report + log + hooks of the bond update in flight
                if (!fl.on) return;
                fl.on = false;
                const long sw = fl.sw; const int b = fl.b, ha = fl.ha;
                tnml_bond_report rep;
                CK(ctx, tnml_bond_update_end(ctx, &rep));
                if (root) {
                    const tnml_bond_report& r_ = rep;
                    std::printf("Sweep %ld Half %d Bond %d\n", sw, ha, r_.c);           // :490
                    std::printf("In cgrad, lambda = %.3E\n", fl.lam);                   // :358
                    for (int p = 0; p < r_.cg.npass_done; ++p) {
                        std::printf("  Conj grad pass %d\n", p + 1);                    // :391
                        const bool has_cost = r_.cg.converged ? true : p + 1 < r_.cg.npass_done || r_.cg.npass_done < Npass;
                        if (has_cost && (p + 1 < Npass)) {
                            std::printf("  Cost = %.10f\n", r_.cg.cost[p] / totNtrain); // :429
                            if (r_.cg.converged && p + 1 == r_.cg.npass_done) std::printf("  |r| = %.1E < %.1E, breaking\n", r_.cg.rnorm[p], cconv);   // :434
                            else std::printf("  |r| = %.1E\n", r_.cg.rnorm[p]);         // :439
                        }
                    }
                    std::printf("Sweep %ld Half %d Bond %d\n", sw, ha, r_.c);           // :510
                    std::printf("SVD trunc err = %.2E\n", r_.truncerr);                 // :523
                    std::printf("Original m=%d, New m=%d\n", r_.origm, r_.newm);        // :525
                    std::printf("norm(newB) = %.12g\n", r_.norm_newB);                  // :528
                    std::printf("rank(newB) = %d\n", r_.label_on_B ? 5 : 4);            // :529 (tensor order)
                    std::printf("|B-newB| = %.3E\n", r_.diff_B_newB);                   // :530
                    for (int l = 0; l < 10; ++l) std::printf("  Label l=%d C%d = %.10f\n", l, l, r_.label_cost[l] / totNtrain);   // :334
                    std::printf("  Reg. cost CR = %.10f\n", r_.reg_cost / totNtrain);   // :337
                    std::printf("Percent correct = %.4f%%, # incorrect = %lld/%d\n", r_.ncorrect * 100. / totNtrain,
                                (long long)(totNtrain - r_.ncorrect), totNtrain);       // :341-342
                    std::printf("--> After SVD, Cost = %.10f\n", r_.cost_after_svd / totNtrain);   // :533
                    if (blog) {                                                         // seconds: wall time since the previous report (pipelined: one bond update of GPU time)
                        const auto t_now = std::chrono::steady_clock::now();
                        std::fprintf(blog, "%ld,%d,%d,%.6e,%d,%.17g,%.17g,%lld,%d,%d,%d,%.6e,%.6f\n", sw, ha, r_.c, fl.lam, r_.cg.npass_done, r_.cost_after_svd / totNtrain,
                                     r_.reg_cost / totNtrain, (long long)r_.ncorrect, totNtrain, r_.origm, r_.newm, r_.truncerr, std::chrono::duration<double>(t_now - t_last).count());
                        t_last = t_now;
                    }
                    const int cs = ha == 1 ? b : b + 1, prevc = ha == 1 ? b - 1 : b + 2;    // :196-209
                    if (prevc >= 1 && prevc <= N) std::printf("## Advancing E from %d to %d\n", prevc, cs);
                    else std::printf("## Making new E at %d\n", cs);
                    // file hooks: rank 0 looks, every rank follows
                    if (file_exists("WRITE_WF")) {                                      // :542-548
                        std::printf("File WRITE_WF found\n");
                        std::remove("WRITE_WF");
                        std::printf("Writing W to disk\n");
                        write_mps("W", download(ctx, N));                               // (pipelined: the network as the bond update in flight leaves it)
                    }
                    if (file_exists("LAMBDA")) {                                        // :550-559
                        std::ifstream lf("LAMBDA"); lf >> lambda_shared; lf.close();
                        std::remove("LAMBDA");
                        std::cout << "new lambda = " << lambda_shared << std::endl;
                    }
                    if (pause_step) { std::printf("PAUSE"); std::fflush(stdout); std::getchar(); }   // :561
                    std::fflush(stdout);
                }
                if (nranks > 1) bar.wait();
                lam = lambda_shared;
                if (nranks > 1) bar.wait();                                             // nobody re-enters the hooks before everyone has read lambda
            };
            for (long sw = 1; sw <= Nsweep; ++sw) {                                         // mldmrg, :470
                if (root) std::printf("\nSweep %ld maxm=%ld minm=%ld\n", sw, maxm, minm);  // :472
                for (int b = 1, ha = 1; ha <= 2; tnml_sweepnext(&b, &ha, N)) {              // :478
                    tnml_sweep_params sp{(int)std::min<long>(maxm, cfg.maxm), (int)std::min<long>(minm, cfg.maxm), cutoff, (int)Npass, lam, lambda_cost, cconv, 0};
                    const InFlight next{sw, b, ha, lam, true};
                    CK(ctx, tnml_bond_update_begin(ctx, b, ha, &sp));
                    finish();                                                               // the previous bond update (none at the start of a sweep)
                    fl = next;
                    if (!pipeline || pause_step) finish();
                }
                finish();
                if (root) {
                    std::printf("Writing W to disk\n");                                     // :565
                    write_mps("W", download(ctx, N));                                       // :566
                }
            }
            if (root) {
                std::printf("Writing W to disk\n");                                         // :763
                write_mps("W", download(ctx, N));                                           // :764
            }
            if (blog) std::fclose(blog);
            if (nranks > 1) { CK(ctx, tnml_replica_check(ctx, nullptr)); bar.wait(); }
            tnml_destroy(ctx);
        };
        (void)write_wf;
        std::vector<std::thread> workers;
        for (int r = 1; r < nranks; ++r) workers.emplace_back(rank_main, r);
        rank_main(0);
        for (auto& t : workers) t.join();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
