"""CPU tests of the drop-in boundary: libtnml.so loads without a GPU, exports every symbol that
include/tnml.h declares, fails loudly instead of falling back, and its host-side rules agree with
the oracle.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tnml.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tnml_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tnml_amd import lib
    L = lib.load()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == names, "tnml_amd.lib.EXPORTS out of sync with include/tnml.h"


def test_create_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tnml_amd.fixedl import TnmlError, TrainStates
    with pytest.raises(TnmlError, match="no HIP device|no CPU fallback"):
        TrainStates(np.zeros(10, dtype=np.int32), 8, 4, pixels=np.zeros((10, 8), dtype=np.uint8))


def test_product_package_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline() may touch oracle/"""
    pat = r"^\s*(from|import)\s+oracle\b|fixedl_oracle|single_oracle|libfixedl_oracle"
    bad = []
    for top in ("tnml_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".cc", ".sh")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(pat, txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    # bench.py: the only use is inside cpu_baseline(); __graft_entry__.py: inside smoke() (and build() compiles it)
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(pat, src, flags=re.M)]
    lo = src.index("def cpu_baseline("); hi = src.index("\ndef ", lo + 1)
    assert uses and all(lo < u < hi for u in uses)


@pytest.mark.parametrize("seed", range(5))
def test_truncate_rule_matches_oracle(seed):
    from oracle import pyoracle
    from tnml_amd import lib
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 40))
    p = np.sort(rng.random(n) ** 8)[::-1].copy()
    p[rng.integers(0, n):] *= 1e-14
    p = np.sort(p)[::-1].copy()
    for maxm, minm, cutoff in [(10, 1, 1e-10), (100, 5, 1e-10), (6, 3, 1e-3), (50, 1, 0.0)]:
        assert lib.truncate(p, maxm, minm, cutoff) == pytest.approx(pyoracle.truncate(p, maxm, minm, cutoff), rel=1e-15, abs=0)


def test_sweepnext_and_shards():
    from oracle import pyoracle
    from tnml_amd import lib
    for N in (4, 9, 784):
        b, ha, seq = 1, 1, []
        while ha <= 2:
            seq.append((b, ha))
            assert lib.sweepnext(b, ha, N) == pyoracle.sweepnext(b, ha, N)
            b, ha = lib.sweepnext(b, ha, N)
        assert len(seq) == 2 * (N - 1)                         # bond updates per sweep
    # ParallelDo chunking (paralleldo.h:32-43): equal chunks, last takes the remainder
    assert [lib.shard_bounds(60000, 8, r) for r in range(8)] == [(7500 * r, 7500 * (r + 1)) for r in range(8)]
    assert [lib.shard_bounds(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]


def test_oneshot_region_bytes_is_what_the_planner_has_to_subtract():
    """tnml_oneshot_region_bytes: [2 parities][nranks][10 Kmax^2 + 48] doubles + flags -- the receive region of the cross-process
    one-shot all-reduce, which tnml_estimate_bytes does not count (no GPU needed: pure arithmetic)"""
    import ctypes as C
    from tnml_amd import lib
    L = lib.load()
    def cfg(maxm, nranks):
        c = lib.Config()
        c.device, c.rank, c.nranks, c.N, c.NT_local, c.maxm, c.dtype = 0, 0, nranks, 784, 7500, maxm, lib.DTYPES["f64"] if hasattr(lib, "DTYPES") else 1
        return c
    b120 = L.tnml_oneshot_region_bytes(C.byref(cfg(120, 8)))
    b300 = L.tnml_oneshot_region_bytes(C.byref(cfg(300, 8)))
    cap120 = 10 * 240 * 240 + 48
    assert 2 * 8 * cap120 * 8 <= b120 <= 2 * 8 * cap120 * 8 + 65536
    assert 4.4e8 < b300 < 4.8e8                                       # ~460 MB per rank at maxm = 300 with 8 ranks
    assert L.tnml_oneshot_region_bytes(None) == -1
