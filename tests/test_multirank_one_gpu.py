"""The library's multi-rank path (image shards, packed [G | cost | ncorrect | pAp] all-reduce, collective truncation
decision, replica fingerprints -- tnml_abi.hip: grad_eval / cgrad_device / quadcost_launch / tnml_bond_update, svd.hip)
with MORE THAN ONE RANK on the one GPU a test box has: every rank gets its own context and host thread, the ranks are
joined by the in-process communicator (tnml_comm_init_local; RCCL itself refuses two ranks on one device and is covered
by test_rccl_path_with_a_one_rank_communicator).  Replaces paralleldo.h:21-68 + the stdx::accumulate reductions
fixedL.cc:333,339,385,402,421,427; the results must not depend on the number of ranks beyond summation order."""
import os
import re
import subprocess
import threading

import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(nranks, labels, phi, W, N, maxm, body, oneshot=False):
    """one context + one host thread per rank on device 0; returns body(ts, rank) of every rank"""
    from tnml_amd import lib
    from tnml_amd.fixedl import TrainStates
    NT = len(labels)
    states = []
    for r in range(nranks):
        lo, hi = lib.shard_bounds(NT, nranks, r)
        states.append(TrainStates(labels[lo:hi], N, maxm, phi=phi[lo:hi], rank=r, nranks=nranks, NT_total=NT))
    if nranks > 1:
        TrainStates.comm_init_local(states, oneshot=oneshot)
        assert all(ts.collective_mode() == (3 if oneshot else 2) for ts in states)
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            ts = states[r]
            ts.set_mps(W)
            out[r] = body(ts, r)
        except Exception as e:                                   # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in th), "a rank hung"
    for e in err:
        if e is not None:
            raise e
    for ts in states:
        ts.close()
    return out


@pytest.mark.parametrize("nranks,NT,oneshot", [(2, 151, False), (3, 150, False), (4, 257, False), (2, 151, True), (4, 257, True)])
def test_sweep_on_several_ranks_matches_one_rank_and_the_oracle(nranks, NT, oneshot):
    from oracle import pyoracle
    from tnml_amd.fixedl import mldmrg
    N, m = 12, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    args = (1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)

    def body(ts, r):
        n = ts.replica_check()
        ts.init()
        B1 = ts.bond_tensor(1)
        G = ts.gradient(B1)                                      # collective: the sum over ALL ranks' images
        C0 = ts.quadcost(B1, 1e-3)
        reps = mldmrg(ts, *args)
        ts.replica_check()
        return dict(n=n, G=G, C0=C0, reps=reps, W=ts.get_mps())
    multi = _run_ranks(nranks, labels, phi, W, N, m, body, oneshot=oneshot)      # oneshot: the peer-write all-reduce (tnml_comm_init_oneshot)
    single = _run_ranks(1, labels, phi, W, N, m, body)[0]
    o = pyoracle.Oracle(phi, labels, W)
    o.init()
    ro = o.mldmrg(*args)
    assert all(x["n"] == nranks for x in multi)
    for x in multi[1:]:                                          # every rank holds the same bits
        assert np.array_equal(x["G"], multi[0]["G"])
        assert [r["cost"] for r in x["reps"]] == [r["cost"] for r in multi[0]["reps"]]
        assert all(np.array_equal(a, b) for a, b in zip(x["W"], multi[0]["W"]))
    x = multi[0]
    assert np.abs(x["G"] - single["G"]).max() <= 1e-12 * np.abs(single["G"]).max()
    assert x["C0"][0] == pytest.approx(single["C0"][0], rel=1e-13) and x["C0"][3] == single["C0"][3]
    assert [r["newm"] for r in x["reps"]] == [r["newm"] for r in single["reps"]] == [r["newm"] for r in ro]
    assert [r["ncorrect"] for r in x["reps"]] == [r["ncorrect"] for r in ro]
    # a different image partition is a different summation order: the same 1e-8 as against the oracle (4 ranks: 1.8e-9 seen)
    np.testing.assert_allclose([r["cost"] for r in x["reps"]], [r["cost"] for r in single["reps"]], rtol=1e-8)
    np.testing.assert_allclose([r["cost"] for r in x["reps"]], [r["cost"] for r in ro], rtol=1e-8)
    np.testing.assert_allclose(np.stack([r["label_cost"] for r in x["reps"]]), np.stack([r["label_cost"] for r in ro]),
                               rtol=1e-7, atol=1e-8 * ro[0]["cost"])


@pytest.mark.parametrize("nranks,oneshot,defer_tail", [(2, False, 1), (3, True, 1), (2, True, 0)])
def test_roll_back_of_a_speculative_split_on_several_ranks(nranks, oneshot, defer_tail):
    """minm = maxm: the split runs speculatively on every rank and the verdict of its deferred check travels in the carried tail
    (slot TNML_SPECSLOT, summed over the ranks), so that every rank rolls back together.  With the test hook debug_fail_split the first,
    an interior and the last speculative split of a pipelined sweep report a failed check on EVERY rank, and -- the case the sum is
    there for -- on rank 1 ALONE: all ranks must repeat the bond update (and the one begun after it) with the synchronous split and
    its collectives, end with bit-identical site tensors, and follow the undisturbed run (tnml_abi.hip: tnml_bond_update_end)."""
    from tnml_amd.fixedl import mldmrg
    N, NT, m = 12, 151, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 5, pixel_boost=200.0)
    args = (1, m, m, 1e-10, 3, 1e-3, 1e-10)

    def body(fail, only_rank):
        def run(ts, r):
            ts.set_option("defer_tail", defer_tail)
            if fail is not None and (only_rank is None or r == only_rank):
                ts.set_option("debug_fail_split", fail)
            ts.init()
            reps = mldmrg(ts, *args, pipelined=True)
            ts.replica_check()
            return dict(reps=reps, W=ts.get_mps(), st=ts.svd_stats())
        return run
    clean = _run_ranks(nranks, labels, phi, W, N, m, body(None, None), oneshot=oneshot)
    for fail, only_rank in ((0, None), (4, None), (17, None), (5, 1)):
        redo = _run_ranks(nranks, labels, phi, W, N, m, body(fail, only_rank), oneshot=oneshot)
        for x in redo:
            assert x["st"]["fallbacks"] >= clean[0]["st"]["fallbacks"] + 1, (fail, only_rank, x["st"], clean[0]["st"])   # EVERY rank repeated the bond update
            assert [r["cost"] for r in x["reps"]] == [r["cost"] for r in redo[0]["reps"]]
            assert all(np.array_equal(a, b) for a, b in zip(x["W"], redo[0]["W"]))
            assert [r["bond"] for r in x["reps"]] == [r["bond"] for r in clean[0]["reps"]]
            assert [r["newm"] for r in x["reps"]] == [r["newm"] for r in clean[0]["reps"]]
            assert [r["ncorrect"] for r in x["reps"]] == [r["ncorrect"] for r in clean[0]["reps"]]
        np.testing.assert_allclose([r["cost"] for r in redo[0]["reps"]], [r["cost"] for r in clean[0]["reps"]], rtol=1e-8)


def test_two_ranks_at_m120_share_the_truncation_decision():
    """m = 120: the in-house eigensolver path with the eigenvalues broadcast from rank 0; two ranks of 300 images"""
    N, NT, m = 20, 600, 120
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)

    def body(ts, r):
        ts.init()
        for bb in range(1, 8):
            ts.shiftE(bb, True)
        reps = [ts.bond_update(b, 1, m, m // 2, 1e-10, 3, 1e-3, 1e-10) for b in (8, 9, 10)]   # Label on RE, on B, on B
        ts.replica_check()
        return reps
    multi = _run_ranks(2, labels, phi, W, N, m, body)
    single = _run_ranks(1, labels, phi, W, N, m, body)[0]
    assert [r["cost"] for r in multi[0]] == [r["cost"] for r in multi[1]]
    assert [r["newm"] for r in multi[0]] == [r["newm"] for r in single]
    assert [r["ncorrect"] for r in multi[0]] == [r["ncorrect"] for r in single]
    np.testing.assert_allclose([r["cost"] for r in multi[0]], [r["cost"] for r in single], rtol=1e-9)


def test_cpp_driver_with_two_ranks_prints_the_one_rank_log(tmp_path):
    """`fixedL` with ngpu = 2 (one host thread + one context per rank; share_device = yes puts both on the one GPU):
    same log lines as the 1-rank run, costs equal up to the summation order of the image shards"""
    from tnml_amd import synth
    N, per_label = 16, 30
    labels = synth.synthetic_labels(10 * per_label, seed=5, per_label=per_label)
    pixels = synth.synthetic_images(N, labels, seed=5)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels[np.argsort(labels, kind="stable")], np.sort(labels), side=4)
    logs = {}

    def costs_of(s):
        return [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", s)]
    for tag, extra in (("one", ""), ("two", "ngpu = 2\nshare_device = yes\n"), ("shot", "ngpu = 2\nshare_device = yes\nallreduce = oneshot\n")):
        wd = tmp_path / tag
        wd.mkdir()
        inp = wd / "input"
        inp.write_text("input\n{\ndatadir = %s\nfeature_scale = 255\nNtrain = %d\nNbatch = 10\nNsweep = 1\ncutoff = 1E-10\nmaxm = 6\n"
                       "minm = 3\nninitial = 4\nlambda = 1E-3\nNpass = 3\nseed = 3\n%s}\n" % (data, per_label, extra))
        run = subprocess.run([os.path.join(ROOT, "tnml_amd", "fixedL"), str(inp)], capture_output=True, text=True, cwd=wd, timeout=600)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
        logs[tag] = run.stdout
    assert "in-process communicator of 2 ranks" in logs["two"] and "one-shot peer-write communicator of 2 ranks" in logs["shot"]
    assert costs_of(logs["shot"]) == costs_of(logs["two"])        # both sum the two ranks' buffers in rank order: the same bits
    assert "Thread 1 150 -> 300 (150)" in logs["two"]

    def costs(s):
        return np.array([float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", s)])

    def skeleton(s):                                             # the log with the numbers blanked
        keep = [ln for ln in s.splitlines() if not ln.startswith(("Thread ", "in-process")) and "communicator" not in ln]
        return [re.sub(r"[-+]?[0-9]*\.?[0-9]+([eE][-+]?[0-9]+)?", "#", ln) for ln in keep]
    c1, c2 = costs(logs["one"]), costs(logs["two"])
    assert len(c1) == 2 * (N - 1) and len(c2) == len(c1)
    np.testing.assert_allclose(c2, c1, rtol=1e-6)
    assert skeleton(logs["one"]) == skeleton(logs["two"])
    assert re.findall(r"New m=(\d+)", logs["one"]) == re.findall(r"New m=(\d+)", logs["two"])


def test_a_diverging_replica_is_detected_and_can_be_repaired():
    """one rank's copy of W.A(b) is moved by one ulp after every split (test hook): with check_replicas = 1 the bond update
    fails on every rank, with check_replicas = 2 rank 0's tensors are re-broadcast, the sweep finishes with identical
    replicas and the costs of the undisturbed run"""
    from tnml_amd.fixedl import TnmlError, mldmrg
    N, NT, m = 12, 150, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    args = (1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)

    def body(mode, nudge):
        def run(ts, r):
            ts.set_option("check_replicas", mode)
            ts.set_option("debug_nudge_rank", nudge)
            ts.init()
            try:
                reps = mldmrg(ts, *args)
            except TnmlError as e:
                return dict(error=str(e))
            ts.replica_check()
            return dict(cost=[x["cost"] for x in reps], repairs=ts.replica_repairs())
        return run
    clean = _run_ranks(2, labels, phi, W, N, m, body(1, -1))
    assert all("error" not in x and x["repairs"] == 0 for x in clean)
    strict = _run_ranks(2, labels, phi, W, N, m, body(1, 1))
    assert all("replicas of W.A" in x.get("error", "") for x in strict)
    repaired = _run_ranks(2, labels, phi, W, N, m, body(2, 1))
    assert all("error" not in x for x in repaired)
    assert repaired[0]["repairs"] == repaired[1]["repairs"] == 2 * (N - 1)
    assert repaired[0]["cost"] == repaired[1]["cost"]
    np.testing.assert_allclose(repaired[0]["cost"], clean[0]["cost"], rtol=1e-9)


def test_a_pipelined_bond_update_enters_five_payload_allreduces():
    """SURVEY.md 8(e) counts 2 Npass + 1 = 9 all-reduces per bond update for the literal cgrad (fixedL.cc:385,402,421 per pass, :333
    after the split); round 2 added a fingerprint max-reduce.  With the merged CG (A p rides with sum |p.v_n|^2) and the after-SVD
    cost partials + fingerprint pieces carried into the next bond update's first all-reduce, a pipelined sweep enters
    1 + (Npass - 1) + 1 = 5 sum all-reduces per bond update (+ the broadcast of rank 0's eigenvalues), and the costs are those of
    the unmerged, undeferred run to round-off."""
    from tnml_amd.fixedl import mldmrg
    N, NT, m, npass = 12, 150, 6, 4
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    args = (1, m, m // 2, 1e-10, npass, 1e-3, 1e-10)

    def body(merged, defer, pipelined):
        def run(ts, r):
            ts.set_option("merged_cg", merged)
            ts.set_option("defer_tail", defer)
            ts.init()
            a0, b0 = ts.collective_stats()
            reps = mldmrg(ts, *args, pipelined=pipelined)
            a1, b1 = ts.collective_stats()
            ts.replica_check()
            return dict(cost=[x["cost"] for x in reps], nc=[x["ncorrect"] for x in reps], newm=[x["newm"] for x in reps],
                        allreduces=a1 - a0, bcasts=b1 - b0, cg=[x["cg"]["cost"][:npass - 1] for x in reps])
        return run
    nb = 2 * (N - 1)
    new = _run_ranks(2, labels, phi, W, N, m, body(1, 1, True))
    old = _run_ranks(2, labels, phi, W, N, m, body(0, 0, False))
    assert new[0]["cost"] == new[1]["cost"]
    assert new[0]["allreduces"] == (npass + 1) * nb + 1, new[0]["allreduces"]     # 5 per bond update + the flush of the last report
    assert old[0]["allreduces"] == (2 * npass + 1) * nb, old[0]["allreduces"]     # 9 per bond update (the fingerprint no longer has its own)
    assert new[0]["bcasts"] == old[0]["bcasts"] == nb
    assert new[0]["newm"] == old[0]["newm"] and new[0]["nc"] == old[0]["nc"]
    # same algebra, different rounding: identical to 1e-9 while the two runs are still on the same trajectory (a free-running sweep
    # amplifies any rounding difference -- the oracle with 1 and with 8 threads does the same), close after
    np.testing.assert_allclose(new[0]["cost"][:4], old[0]["cost"][:4], rtol=1e-9)
    # bonds 5 and 6 carry the Label index on B: there the merged recurrence and the literal order differ in the later step sizes at 1e-3
    # and in the cost at ~1e-9 (tnml_abi.hip, cgrad_device; 1.7e-9 at bond 6 with the round-5 split kernels, 0.9e-9 with round 4's)
    np.testing.assert_allclose(new[0]["cost"][:8], old[0]["cost"][:8], rtol=1e-8)
    np.testing.assert_allclose(new[0]["cost"], old[0]["cost"], rtol=1e-3)
    for a, b in zip(new[0]["cg"][:8], old[0]["cg"][:8]):                          # the per-pass costs the reference prints (:429)
        np.testing.assert_allclose(a, b, rtol=1e-9)


def test_split_on_the_workgroup_cluster_keeps_the_replicas_identical():
    """bond dimension 150 (Gram side > 240): every rank runs its own multi-workgroup tridiagonalisation (eigh_mc.hip) -- the partial
    sums are exchanged through tagged granules and summed in workgroup order, so two ranks (two concurrent clusters on one GPU)
    must produce bit-identical site tensors; checked by the fingerprint of every bond update and by tnml_replica_check"""
    from tnml_amd.fixedl import mldmrg
    N, NT, m = 20, 48, 150
    pixels, labels, phi, W = make_problem(N, NT, m, 9, pixel_boost=200.0)
    args = (1, m, m // 2, 1e-10, 2, 1e-3, 1e-10)

    def body(ts, r):
        ts.init()
        reps = mldmrg(ts, *args, max_bonds=11, pipelined=True)
        ts.replica_check()
        return dict(cost=[x["cost"] for x in reps], newm=[x["newm"] for x in reps], fb=ts.svd_stats()["fallbacks"])
    two = _run_ranks(2, labels, phi, W, N, m, body)
    one = _run_ranks(1, labels, phi, W, N, m, body)[0]
    assert two[0]["cost"] == two[1]["cost"] and two[0]["newm"] == two[1]["newm"] == one["newm"]
    assert max(one["newm"]) == 150 and two[0]["fb"] == 0 and one["fb"] == 0
    np.testing.assert_allclose(two[0]["cost"], one["cost"], rtol=1e-7)


def test_a_cluster_that_gives_up_on_one_rank_sends_every_rank_to_the_fallback():
    """svd.hip: when the workgroup cluster of ONE rank gives up (mc_spin_max = 0 on rank 1 only), the redo with rocsolver_dsyevd contains
    a broadcast -- every rank has to take it, on the summed status words, or the collective sequences of the ranks part ways (round-3
    advisor finding).  Both ranks must report the same fallbacks, bit-identical tensors and the one-rank costs."""
    from tnml_amd.fixedl import mldmrg
    N, NT, m = 20, 48, 150
    pixels, labels, phi, W = make_problem(N, NT, m, 9, pixel_boost=200.0)
    args = (1, m, m // 2, 1e-10, 2, 1e-3, 1e-10)

    def body(ts, r, spin=True):
        if spin and r == 1:
            ts.set_option("mc_spin_max", 0)
        ts.init()
        reps = mldmrg(ts, *args, max_bonds=11, pipelined=True)
        ts.replica_check()
        return dict(cost=[x["cost"] for x in reps], newm=[x["newm"] for x in reps], fb=ts.svd_stats()["fallbacks"])
    two = _run_ranks(2, labels, phi, W, N, m, body)
    one = _run_ranks(1, labels, phi, W, N, m, lambda ts, r: body(ts, r, False))[0]
    assert two[0]["fb"] == two[1]["fb"] > 0 and one["fb"] == 0
    assert two[0]["cost"] == two[1]["cost"] and two[0]["newm"] == two[1]["newm"] == one["newm"]
    np.testing.assert_allclose(two[0]["cost"], one["cost"], rtol=1e-7)


@pytest.mark.parametrize("noise", [0.0, 1e-5])
def test_per_label_variant_on_two_ranks_with_and_without_noise(noise):
    """TNML_MODE_SINGLE with the images sharded over two ranks: the sums of the CG and -- with noise > 0 -- the images' term of the
    density-matrix split (three weighted Gram matrices of the environment, all-reduced before they are added to rho, single.h:654-665)
    are sums over ranks; costs and kept dimensions must match the one-rank run and the oracle, the replicas must stay bit-identical."""
    from oracle import pyoracle
    from tnml_amd import lib, synth
    from tnml_amd.fixedl import TrainStates, mldmrg
    N, NT, m, target = 10, 90, 3, 7
    labels = synth.synthetic_labels(NT, seed=4, per_label=NT // 10)
    pixels = synth.synthetic_images(N, labels, seed=4)
    phi = pyoracle.features_single(pixels, True).copy()
    phi[..., 1] *= 300.0
    W = synth.random_mps(N, m, seed=11)
    W[N // 2 - 1] = W[N // 2 - 1][..., 0] * 3.0
    o = pyoracle.SingleOracle(phi, labels, target, W)
    o.set_noise(noise)
    o.init()
    ro = o.mldmrg(1, 5, 2, 1e-10, 3, 1e-3, 1e-10)

    def run(nranks):
        states = []
        for r in range(nranks):
            lo, hi = lib.shard_bounds(NT, nranks, r)
            states.append(TrainStates(labels[lo:hi], N, 5, phi=phi[lo:hi], rank=r, nranks=nranks, NT_total=NT, single_label=target))
        if nranks > 1:
            TrainStates.comm_init_local(states)
        out, err = [None] * nranks, [None] * nranks

        def work(r):
            try:
                ts = states[r]
                ts.set_mps(W)
                if noise:
                    ts.set_option_real("noise", noise)
                ts.init()
                reps = mldmrg(ts, 1, 5, 2, 1e-10, 3, 1e-3, 1e-10)
                out[r] = (reps, [ts.get_site(j) for j in range(1, N + 1)])
            except Exception as e:                               # noqa: BLE001
                err[r] = e
        th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in th), "a rank hung"
        for e in err:
            if e is not None:
                raise e
        for ts in states:
            ts.close()
        return out
    one, two = run(1), run(2)
    for a, b, c in zip(two[0][0], one[0][0], ro):
        assert a["newm"] == b["newm"] == c["newm"]
        assert a["cost"] == pytest.approx(b["cost"], rel=1e-9) and a["cost"] == pytest.approx(c["cost"], rel=1e-7)
        assert a["truncerr"] == pytest.approx(c["truncerr"], rel=1e-3, abs=1e-12)
    for A0, A1 in zip(two[0][1], two[1][1]):
        assert np.array_equal(A0, A1)                              # replicas bit-identical


def test_a_local_error_leaves_the_in_process_communicator_usable_and_a_failed_collective_does_not():
    """tnml_fail aborts an in-process communicator only from inside an entry point that every rank calls in step (round-3 advisor
    finding: a bad option name on one rank used to poison every later collective).  First half: rank 1 makes a benign mistake
    (unknown option), then both ranks sweep and agree.  Second half: rank 1 fails INSIDE a collective entry point (a bond update
    with half = 3); rank 0, already waiting in its all-reduce, must return an error within the configured comm_timeout_s
    instead of hanging -- the failure is propagated by the abort, not by the timeout."""
    import time
    from tnml_amd.fixedl import TnmlError, mldmrg
    N, NT, m = 8, 64, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 4)
    args = (1, m, 1, 1e-10, 2, 1e-3, 1e-10)

    def body(ts, r):
        ts.set_option("comm_timeout_s", 60)
        if r == 1:
            with pytest.raises(TnmlError):
                ts.set_option("no_such_option", 1)
        ts.init()
        reps = mldmrg(ts, *args, max_bonds=5)
        ts.replica_check()
        t0 = time.time()
        with pytest.raises(TnmlError):
            if r == 1:
                ts.bond_update(3, 3, *args[1:])                # rejected by the argument check of a collective entry point
            else:
                ts.init()
                mldmrg(ts, *args, max_bonds=2)                 # waits for rank 1 in its first all-reduce
        return [x["cost"] for x in reps], time.time() - t0
    two = _run_ranks(2, labels, phi, W, N, m, body)
    assert two[0][0] == two[1][0]
    assert two[0][1] < 30.0, "rank 0 was released by the time-out, not by the abort"


def test_ranks_with_an_environment_budget_keep_their_replicas_and_their_results():
    """the host tier of the environments (option env_budget_mb) under the collective path: two ranks on one GPU, each with its own budget of
    one MiB -- every rank evicts and fetches on its own second stream while the all-reduces go on; costs, bond dimensions and site tensors
    are bit-identical to the same two ranks with everything resident, and both ranks really spilled.  The per-label variant (label extent
    1: ten environments share a slab) under a budget, on one rank, against its resident run."""
    from tnml_amd.fixedl import TrainStates, mldmrg
    N, NT, m = 25, 180, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 5, pixel_boost=200.0)
    args = (1, m, m // 2, 1e-10, 3, 1e-3, 1e-10)

    def body(budget):
        def run(ts, r):
            if budget:
                ts.set_option("env_budget_mb", 1)                # 8 slabs of 10 x 6 x 256 doubles
            ts.init()
            reps = mldmrg(ts, *args, pipelined=True)
            ts.replica_check()
            return dict(cost=[x["cost"] for x in reps], newm=[x["newm"] for x in reps], W=ts.get_mps(), st=ts.env_stats())
        return run
    tight = _run_ranks(2, labels, phi, W, N, m, body(True))
    free = _run_ranks(2, labels, phi, W, N, m, body(False))
    for r in range(2):
        assert tight[r]["st"]["spills"] > 10 and tight[r]["st"]["fetches"] > 5 and free[r]["st"]["spills"] == 0
        assert tight[r]["cost"] == free[r]["cost"] == tight[0]["cost"] and tight[r]["newm"] == free[r]["newm"]
        assert all(np.array_equal(a, b) for a, b in zip(tight[r]["W"], free[r]["W"]))
    # per-label variant
    Ws = [A[..., 0] if A.ndim == 4 else A for A in W]
    out = []
    for budget in (0, 1):
        ts = TrainStates(labels, N, m, phi=phi, single_label=3)
        if budget:
            ts.set_option("env_budget_mb", budget)
        ts.set_mps(Ws)
        ts.init()
        reps = mldmrg(ts, 2, m, m // 2, 1e-10, 3, 1e-3, 1e-10)
        out.append(([x["cost"] for x in reps], ts.env_stats()))
        ts.close()
    assert out[0][0] == out[1][0]
    assert out[0][1]["spills"] == 0


def _run_processes(nranks, NT, mode):
    """one PROCESS per rank, all on device 0, joined by the cross-process one-shot all-reduce (tnml_oneshot_export / _connect): the
    parent only carries the IPC handles between them"""
    import json
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker = os.path.join(ROOT, "tests", "mp_oneshot_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(nranks), str(NT), mode], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(nranks)]
    try:
        handles = []
        for p in procs:
            line = p.stdout.readline()
            while line and not line.startswith("HANDLE "):
                line = p.stdout.readline()
            assert line.startswith("HANDLE "), p.stderr.read()[-3000:]
            handles.append(line.split()[1])
        for p in procs:
            p.stdin.write(" ".join(handles) + "\n")
            p.stdin.flush()
        outs = []
        for p in procs:
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0, se[-3000:]
            res = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
            assert res, (so[-2000:], se[-2000:])
            outs.append(json.loads(res[-1][7:]))
        return outs
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.parametrize("nranks,NT,mode", [(2, 151, "truncating"), (2, 151, "spec"), (3, 150, "truncating"), (3, 151, "spec")])
def test_one_shot_allreduce_across_processes(nranks, NT, mode):
    """the one-shot all-reduce as `bench.py --gpus N --allreduce oneshot` reaches it -- one process per rank,
    receive regions mapped through hipIpcOpenMemHandle, device-side arrival flags, no host barrier (ipc_comm.hip).  Two / three
    processes on the one GPU: the sweep must give every rank the same bits, and the same bits as the in-process one-shot form with
    the same image shards (both sum the shards in rank order).  mode "spec": minm = maxm, i.e. the speculative split and its carried
    check flag run across the ranks as well; "truncating": the eigenvalue broadcast is one of the collectives."""
    from tnml_amd.fixedl import mldmrg
    N, m = 12, 6
    pixels, labels, phi, W = make_problem(N, NT, m, 3, pixel_boost=200.0)
    minm = m if mode == "spec" else m // 2
    outs = _run_processes(nranks, NT, mode)

    def body(ts, r):
        ts.init()
        B1 = ts.bond_tensor(1)
        G = ts.gradient(B1)
        reps = mldmrg(ts, 1, m, minm, 1e-10, 3, 1e-3, 1e-10, pipelined=True)
        return dict(G=G, reps=reps, W=ts.get_mps())
    inproc = _run_ranks(nranks, labels, phi, W, N, m, body, oneshot=True)
    single = _run_ranks(1, labels, phi, W, N, m, body)[0]
    assert all(o["n"] == nranks for o in outs)
    for o in outs:
        assert o["cost"] == outs[0]["cost"] and o["newm"] == outs[0]["newm"] and o["W"] == outs[0]["W"]      # every rank: the same bits
        assert np.array_equal(np.asarray(o["G"]).reshape(inproc[0]["G"].shape), inproc[0]["G"])           # = the in-process form
        assert o["cost"] == [r["cost"] for r in inproc[0]["reps"]]
        assert o["newm"] == [r["newm"] for r in inproc[0]["reps"]] == [r["newm"] for r in single["reps"]]
        assert o["ncorrect"] == [r["ncorrect"] for r in single["reps"]]
    np.testing.assert_allclose(outs[0]["cost"], [r["cost"] for r in single["reps"]], rtol=1e-8)            # another image partition: summation order
    nb = 2 * (N - 1)
    if mode == "spec":
        assert outs[0]["bcasts"] <= 4, outs[0]["bcasts"]         # only the four 2 x 2 chain-end splits still broadcast eigenvalues
    else:
        assert outs[0]["bcasts"] == nb


@pytest.mark.parametrize("nranks", [2, 3])
def test_one_shot_allreduce_across_processes_at_m120(nranks):
    """The cross-process one-shot all-reduce at the HEADLINE payload: [48 scalars | G], G = 240 x 240 doubles = 461 KB = 29 chunks of
    k_os_exchange (the m = 6 test above moves one chunk).  Two and three PROCESSES on the one GPU, m = 120, pipelined bond updates with
    the speculative split, strict replica check (a mismatch is an error); the parent holds the ranks at a start barrier until every one
    of them has its context, data and peers (tools/oneshot_processes_m120.py).  Every rank must hold the same bits: six all-reduced
    gradients, the cost of each of 8 bond updates, every site tensor at the end."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oneshot_processes_m120 as osp
    outs, errs = osp.run(nranks, NT=1536, nbonds=8, repair=0, timeout=420)
    osp.check(outs, nranks)
    assert all(o["mem_kind"] in (1, 2) for o in outs)
    assert outs[0]["allreduces"] >= 6 + 5 * 8
