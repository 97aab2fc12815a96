"""The cross-process one-shot all-reduce (ipc_comm.hip) at the headline payload between PROCESSES sharing the one GPU of a box:
starts tests/mp_oneshot_m120_worker.py once per rank, carries the IPC handles, holds the ranks at a start barrier, compares the ranks'
bits and prints every rank's timeline.  `run()` is what the -m gpu test calls.
  python tools/oneshot_processes_m120.py [nranks=3] [NT=1536] [nbonds=8] [repeats=1]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(nranks, NT=1536, nbonds=8, repair=0, timeout=420, verbose=False):
    """returns (results per rank, stderr timelines per rank); raises AssertionError with the ranks' stderr on any failure"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker = os.path.join(ROOT, "tests", "mp_oneshot_m120_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(nranks), str(NT), str(nbonds), str(repair)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(nranks)]

    def expect(p, prefix):
        line = p.stdout.readline()
        while line and not line.startswith(prefix):
            line = p.stdout.readline()
        assert line.startswith(prefix), "a rank ended before '%s': %s" % (prefix, p.stderr.read()[-3000:])
        return line
    try:
        handles = [expect(p, "HANDLE ").split()[1] for p in procs]
        for p in procs:
            p.stdin.write(" ".join(handles) + "\n")
            p.stdin.flush()
        for p in procs:
            expect(p, "READY")
        for p in procs:                                            # the start barrier: every rank has its context, data and peers
            p.stdin.write("GO\n")
            p.stdin.flush()
        outs, errs, bad = [], [], []
        t_end = time.time() + timeout
        for r, p in enumerate(procs):
            so, se = p.communicate(timeout=max(1.0, t_end - time.time()))
            errs.append(se)
            res = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
            if p.returncode != 0 or not res:
                bad.append("rank %d failed (rc %s):\n%s" % (r, p.returncode, se[-2500:]))
            else:
                outs.append(json.loads(res[-1][7:]))
        assert not bad, "\n".join(bad)
        if verbose:
            for se in errs:
                sys.stderr.write(se)
        return outs, errs
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def check(outs, nranks):
    assert all(o["n"] == nranks for o in outs)
    for o in outs:
        assert o["grad"] == outs[0]["grad"], "all-reduced gradients differ between ranks"
        assert len(set(o["grad"])) == 1, "the all-reduced gradient changes from one evaluation to the next"
        assert o["cost"] == outs[0]["cost"], "per-bond costs differ between ranks"
        assert o["W"] == outs[0]["W"], "site tensors differ between ranks"
        assert o["repairs"] == 0


def main():
    nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    NT = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
    nbonds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    ok = 0
    for i in range(reps):
        t0 = time.time()
        try:
            outs, errs = run(nranks, NT, nbonds, verbose=(i == 0))
            check(outs, nranks)
            ok += 1
            print("run %d: %d processes, %d bond updates at m = 120: every rank holds the same bits (%.1f s; all-reduces %d, broadcasts %d, mem_kind %s, svd %s)" %
                  (i, nranks, nbonds, time.time() - t0, outs[0]["allreduces"], outs[0]["bcasts"], outs[0]["mem_kind"], outs[0]["svd"]), flush=True)
        except Exception as e:                                   # noqa: BLE001
            print("run %d FAILED after %.1f s: %s" % (i, time.time() - t0, str(e)[-3000:]), flush=True)
    print("%d of %d runs clean" % (ok, reps))
    return 0 if ok == reps else 1


if __name__ == "__main__":
    sys.exit(main())
