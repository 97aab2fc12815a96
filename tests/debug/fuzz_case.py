"""replay one case of tests/debug/fuzz_lockstep.py (same generator, same seed) and print the CG traces of the bond update with the largest
deviation: HIP path in its default and in the literal evaluation order against the oracle with 1 and with 3 threads
   python tests/debug/fuzz_case.py <seed> <case>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "debug"))
import fuzz_lockstep as fz
from conftest import make_problem
from oracle import pyoracle
from tnml_amd import lib
from tnml_amd.fixedl import TrainStates
from test_gpu_parity import _mps_with_dims
seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
# re-draw the parameters exactly as one_case does
class Stop(Exception): pass
desc = None
orig = fz.make_problem
def grab(N, NT, m, sd, pixel_boost):
    raise Stop((N, NT, sd, pixel_boost))
for idx in range(case + 1):
    st = rng.bit_generator.state
    if idx < case:
        fz.make_problem = lambda *a, **k: (_ for _ in ()).throw(Stop())
        try: fz.one_case(rng, idx)
        except Stop: pass
    else:
        rng.bit_generator.state = st
        N = int(rng.integers(4, 15)); big = rng.random() < 0.15; cap = 160 if big else 40
        dims = [1]
        for j in range(1, N):
            lim = min(cap, 2 * dims[-1], 2 ** min(N - j, 12)); dims.append(int(rng.integers(1, max(2, lim + 1))))
        dims.append(1)
        for j in range(N - 1, 0, -1): dims[j] = min(dims[j], 2 * dims[j + 1])
        NT = int(rng.choice([1, 7, 33, 64, 100, 257])); boost = float(rng.choice([1.0, 30.0, 200.0])); dtype = str(rng.choice(["f64", "f64", "f64", "f64_e32", "f32"]))
        maxm = int(rng.integers(2, max(3, max(dims) + 3))); minm = int(rng.integers(1, maxm + 1)); cutoff = float(rng.choice([0.0, 1e-12, 1e-8, 1e-4]))
        npass = int(rng.integers(1, 5)); lam = float(rng.choice([1e-3, 1e-2, 1e-1]))
print("case", case, dict(N=N, dims=dims, NT=NT, boost=boost, dtype=dtype, maxm=maxm, minm=minm, cutoff=cutoff, npass=npass, lam=lam))
pixels, labels, phi, _ = orig(N, NT, 2, 5 + case, pixel_boost=boost)
W = _mps_with_dims(dims, 100 + case)
for name, opts, nth in (("default order, oracle 1 thread", {}, 1), ("literal order, oracle 3 threads", {"fast_cg": 0, "reuse_p": 0}, 3)):
    ts = TrainStates(labels, N, max(max(dims), maxm), phi=phi, dtype=dtype)
    for k, v in opts.items(): ts.set_option(k, v)
    o = pyoracle.Oracle(phi, labels, W, nthread=nth)
    ts.set_mps(W); o.init(); ts.init()
    b, ha = 1, 1
    print(name)
    while ha <= 2:
        o.set_bond(b); ts.setBond(b)
        B0 = o.bond_tensor(b)
        Bg, tg = ts.cgrad(B0, npass, lam, 1e-10)
        Bo, to = o.cgrad(B0, npass, lam, 1e-10)
        da = max(abs(a / c - 1) for a, c in zip(tg["alpha"], to["alpha"])) if len(to["alpha"]) else 0.
        print("  bond %d half %d: max rel. alpha deviation %.1e  alphas %s | cost(B_gpu) %.10f cost(B_oracle) %.10f" % (b, ha, da, ["%.4g" % x for x in to["alpha"]], o.quadcost(Bg, lam)[0], o.quadcost(Bo, lam)[0]))
        r = ts.bond_update(b, ha, maxm, minm, cutoff, npass, lam, 1e-10)
        o.svd_split(Bo, b, ha, cutoff, maxm, minm); o.shiftE(b, ha == 1)
        ts.set_site(b, o.get_site(b)); ts.set_site(b + 1, o.get_site(b + 1)); ts.shiftE(b, ha == 1)
        b, ha = lib.sweepnext(b, ha, N)
    ts.close()
