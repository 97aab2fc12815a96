import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_problem(N=12, NT=60, m=4, seed=3, s1_scale=1.0, pixel_boost=1.0):
    """Small seeded (pixels, labels, phi, W) problem shared by the oracle and GPU parity tests."""
    from tnml_amd import synth
    labels = synth.synthetic_labels(NT, seed=seed, per_label=NT // 10 if NT % 10 == 0 else None)
    pixels = synth.synthetic_images(N, labels, seed=seed)
    phi = synth.features_series(pixels)
    if pixel_boost != 1.0:          # stronger second feature component: better-conditioned tests
        phi = phi.copy()
        phi[..., 1] *= pixel_boost
    W = synth.random_mps(N, m, seed=seed + 7, s1_scale=s1_scale)
    return pixels, labels, phi, W


@pytest.fixture(scope="session")
def small_problem():
    return make_problem()


def pytest_sessionstart(session):
    """a fresh checkout has no built libraries (they are git-ignored): build them once before the first test needs them"""
    need = [os.path.join(ROOT, "tnml_amd", "libtnml.so"), os.path.join(ROOT, "tnml_amd", "libtnml_host.so"),
            os.path.join(ROOT, "oracle", "libfixedl_oracle.so"), os.path.join(ROOT, "tnml_amd", "fixedL")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
