"""The CPU oracle under sanitizers (the reference has no race detection; its thread safety is by
construction -- per-thread accumulator slots, every image written by its owning chunk -- and the oracle restates that with
pthreads).  A small C harness drives one sweep of the oracle with 3 worker threads on synthetic data, built once with
-fsanitize=address,undefined and once with -fsanitize=thread: any report fails the test.  Test infrastructure only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "fixedl_oracle.h"
static double rnd(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0 - 0.5; }
int main(void) {
    enum { N = 8, NT = 30, M = 3 };
    unsigned seed = 12345u;
    double* phi = malloc(sizeof(double) * NT * N * 2);
    int labels[NT];
    for (int n = 0; n < NT; ++n) { labels[n] = n % 10; for (int j = 0; j < N; ++j) { phi[(n * N + j) * 2] = 1.0; phi[(n * N + j) * 2 + 1] = 0.5 + rnd(&seed) + 0.05 * labels[n]; } }
    orc* o = orc_create(N, NT, phi, labels, 3, 1);
    if (!o) { printf("create failed: %s\n", orc_last_error()); return 2; }
    int dims[N + 1]; dims[0] = 1; dims[N] = 1;
    for (int j = 1; j < N; ++j) { int a = 1 << (j < N - j ? j : N - j); dims[j] = a < M ? a : M; }
    for (int j = 1; j <= N; ++j) {
        const int ml = dims[j - 1], mr = dims[j], L = (j == N / 2) ? 10 : 1;
        double* A = malloc(sizeof(double) * ml * 2 * mr * L);
        for (int i = 0; i < ml * 2 * mr * L; ++i) A[i] = 0.3 * rnd(&seed);
        for (int a = 0; a < ml && a < mr; ++a) for (int l = 0; l < L; ++l) A[a + ml * (0 + 2 * (a + mr * l))] += 1.0 / sqrt((double)L);
        if (orc_set_site(o, j, ml, mr, L == 10, A)) { printf("set_site failed: %s\n", orc_last_error()); return 2; }
        free(A);
    }
    if (orc_init(o)) { printf("init failed: %s\n", orc_last_error()); return 2; }
    if (orc_mldmrg(o, 1, M, 1, 1e-10, 2, 1e-3, 1e-10, 0, NULL, 0) < 0) { printf("mldmrg failed: %s\n", orc_last_error()); return 2; }
    double out[10];
    if (orc_toverlap(o, 0, out)) return 2;
    printf("ok %.6f\n", out[0]);
    orc_destroy(o);
    free(phi);
    return 0;
}
"""


@pytest.mark.parametrize("name,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]), ("tsan", ["-fsanitize=thread"])])
def test_oracle_sweep_is_clean_under_sanitizers(tmp_path, name, flags):
    src = tmp_path / "harness.c"
    src.write_text(HARNESS)
    exe = tmp_path / ("harness_" + name)
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-pthread", "-I", os.path.join(ROOT, "oracle")] + flags + \
          [str(src), os.path.join(ROOT, "oracle", "fixedl_oracle.c"), os.path.join(ROOT, "oracle", "single_oracle.c"), "-o", str(exe), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("this gcc has no %s runtime: %s" % (name, b.stderr[-200:]))
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
