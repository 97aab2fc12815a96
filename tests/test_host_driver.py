"""CPU tests of the host side of the fixedL drop-in (tnml_amd/host): input-file grammar, idx-ubyte
reader with the reference's per-label selection, TNMLW1 weight files, initial-W builder."""
import os
import re
import subprocess

import numpy as np
import pytest

from tnml_amd import hostlib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = os.path.join(ROOT, "tests", "golden", "input_fixedL_sample")


def test_input_file_grammar():
    """same grammar as sample_inputs/input_fixedL: group `input { key = value ... }`, several pairs per
    line, unknown keys ignored, yes/no by first letter, missing keys fall back to defaults"""
    assert hostlib.input_get(SAMPLE, "datadir") == "/data/MNIST"
    assert int(hostlib.input_get(SAMPLE, "Ntrain")) == 100
    assert float(hostlib.input_get(SAMPLE, "cutoff")) == 1e-12
    assert int(hostlib.input_get(SAMPLE, "maxm")) == 40 and int(hostlib.input_get(SAMPLE, "Nsweep")) == 50
    assert int(hostlib.input_get(SAMPLE, "Npass")) == 2
    assert hostlib.input_get(SAMPLE, "imglen") == "28"            # present, never read by fixedL.cc
    assert hostlib.input_get(SAMPLE, "cconv") is None             # absent -> default in the driver
    assert hostlib.input_yesno(SAMPLE, "pause_step") is True
    assert hostlib.input_yesno(SAMPLE, "replace", True) is False
    assert hostlib.input_yesno(SAMPLE, "nokey", True) is True
    with pytest.raises(RuntimeError, match="Couldn't open"):
        hostlib.input_get("/nonexistent/file", "x")


def test_input_file_grammar_of_the_per_label_sample():
    """layout of sample_inputs/input_single: the group braces and the keys indented"""
    f = os.path.join(ROOT, "tests", "golden", "input_single_sample")
    assert hostlib.input_get(f, "datadir") == "/data/MNIST"
    assert int(hostlib.input_get(f, "label")) == 3 and int(hostlib.input_get(f, "imglen")) == 8
    assert int(hostlib.input_get(f, "maxm")) == 20 and float(hostlib.input_get(f, "cutoff")) == 1e-9
    assert float(hostlib.input_get(f, "lambda")) == 1e-8 and hostlib.input_get(f, "feature") is None


def test_driver_usage_and_missing_data(tmp_path):
    exe = os.path.join(ROOT, "tnml_amd", "fixedL")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("Usage:")            # fixedL.cc:579-583
    out = subprocess.run([exe, SAMPLE], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 1 and "Error opening file /data/MNIST/train-images-idx3-ubyte" in out.stderr


def _dataset(tmp_path, N=16, per_label=7, seed=5):
    labels = synth.synthetic_labels(10 * per_label, seed=seed, per_label=per_label)
    pixels = synth.synthetic_images(N, labels, seed=seed)
    d = str(tmp_path / "data")
    synth.write_idx(d, pixels, labels)
    return d, pixels, labels


def test_idx_reader_selection_rule(tmp_path):
    """mllib/mnist.h:472-496: first NT images per label in file order"""
    d, pixels, labels = _dataset(tmp_path)
    px, lab, idx = hostlib.read_mnist(d, True, 3)
    keep, cnt = [], [0] * 10
    for i, l in enumerate(labels):
        if cnt[l] < 3:
            cnt[l] += 1
            keep.append(i)
    assert list(idx) == keep and len(keep) == 30
    np.testing.assert_array_equal(px, pixels[keep])
    np.testing.assert_array_equal(lab, labels[keep])
    px_all, _, _ = hostlib.read_mnist(d, True, 60000)
    assert px_all.shape == pixels.shape
    with open(os.path.join(d, "train-labels-idx1-ubyte"), "r+b") as f:       # corrupt the magic number
        f.write(b"\x00\x00\x08\x02")
    with pytest.raises(RuntimeError, match="magic"):
        hostlib.read_mnist(d, True, 3)


def test_weight_file_roundtrip(tmp_path):
    W = synth.random_mps(12, 5, seed=3)
    f = str(tmp_path / "W")
    hostlib.write_mps(f, W)
    W2 = hostlib.read_mps(f)
    assert len(W2) == 12
    for a, b in zip(W, W2):
        assert a.shape == b.shape
        np.testing.assert_array_equal(a, b)
    with open(f, "r+b") as fh:
        fh.write(b"XXXX")
    with pytest.raises(RuntimeError, match="TNMLW1"):
        hostlib.read_mps(f)


def test_initial_w_builder(tmp_path):
    """fixedL.cc:702-728: Label on site N/2 only, bond dimension <= 10, centre tensor normalised,
    reproducible for a given seed; for ninitial=1 the MPS equals sum_l 0.1 |image_l> (x) e_l exactly"""
    from oracle import pyoracle
    d, pixels, labels = _dataset(tmp_path, N=16, per_label=7)
    f1, f2, f3 = (str(tmp_path / n) for n in ("W1", "W2", "W3"))
    ovl, md = hostlib.build_initial_w(d, 7, 4, 11, f1)
    hostlib.build_initial_w(d, 7, 4, 11, f2)
    hostlib.build_initial_w(d, 7, 4, 12, f3)
    W1, W2, W3 = hostlib.read_mps(f1), hostlib.read_mps(f2), hostlib.read_mps(f3)
    N, c0 = 16, 8
    assert md <= 10 and ovl > 0
    for j, A in enumerate(W1, start=1):
        assert (A.ndim == 4) == (j == c0)
    assert np.linalg.norm(W1[c0 - 1]) == pytest.approx(1.0, rel=1e-12)
    assert all(np.array_equal(a, b) for a, b in zip(W1, W2))
    assert any(a.shape != b.shape or not np.allclose(a, b) for a, b in zip(W1, W3))
    # ninitial = 1: check the model output against the closed form on every image
    f4 = str(tmp_path / "W4")
    hostlib.build_initial_w(d, 7, 1, 3, f4)
    W4 = hostlib.read_mps(f4)
    phi = synth.features_series(pixels)
    o = pyoracle.Oracle(phi, labels, W4)
    out = np.stack([o.toverlap(i) for i in range(len(labels))])          # [n, 10]
    # closed form: out[n, l] = 0.1 <chosen_l | image_n> / norm, with one (unknown) chosen image per label:
    # every column must then be proportional to the overlaps with SOME training image of that label
    G = np.prod(np.einsum("nis,mis->nmi", phi, phi), axis=2)             # <image_n | image_m>
    for l in range(10):
        cands = np.where(labels == l)[0]
        best = min(np.abs(out[:, l] / out[:, l].max() - G[:, m] / G[:, m].max()).max() for m in cands)
        assert best < 1e-5            # the label sum is truncated with Cutoff 1E-8 (fixedL.cc:724)


def test_block_mean_reduce():
    """imglen down-sampling (image.h:316-346 `reduce`): plain mean of side/newlen blocks starting at side % bsize"""
    from tnml_amd import hostlib
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, size=(7, 28 * 28), dtype=np.uint8)
    r = hostlib.reduce(px, 28, 14)
    ref = px.reshape(7, 14, 2, 14, 2).astype(np.float64).mean(axis=(2, 4)).reshape(7, 196)
    np.testing.assert_array_equal(r, ref)                       # sums of 4 bytes / 4: exact in fp64
    # side not divisible by newlen: bsize = 28 // 9 = 3, rem = 28 % 3 = 1 -> blocks start at pixel 1
    r9 = hostlib.reduce(px, 28, 9)
    img = px.reshape(7, 28, 28).astype(np.float64)
    ref9 = np.stack([[img[i, 1 + 3 * y:4 + 3 * y, 1 + 3 * x:4 + 3 * x].mean() for y in range(9) for x in range(9)] for i in range(7)])
    np.testing.assert_allclose(r9, ref9, rtol=1e-15)
    np.testing.assert_array_equal(hostlib.reduce(px, 28, 28), px.astype(np.float64))
    with pytest.raises(RuntimeError):
        hostlib.reduce(px, 28, 29)


def test_initial_w_of_the_per_label_variant(tmp_path):
    """single.cc:112-128: normalised sum of product states of the selected label, orthogonality centre on site 1"""
    from tnml_amd import hostlib, synth
    N, per_label = 16, 12
    labels = synth.synthetic_labels(10 * per_label, seed=3, per_label=per_label)
    pixels = np.clip(synth.synthetic_images(N, labels, seed=3).astype(np.int32) * 3, 0, 255).astype(np.uint8)
    data = str(tmp_path / "data")
    synth.write_idx(data, pixels, labels)
    out = str(tmp_path / "W4")
    hostlib.build_initial_single(data, per_label, 4, 5, 7, True, out)
    W = hostlib.read_mps(out)
    assert len(W) == N and all(A.ndim == 3 for A in W) and max(max(A.shape[0], A.shape[2]) for A in W) <= 10
    # norm 1, and sites 2..N right-orthonormal (centre on site 1)
    E = np.ones((1, 1))
    for A in W:
        E = np.einsum('ab,asr,bsq->rq', E, A, A)
    assert E[0, 0] == pytest.approx(1.0, rel=1e-10)
    for A in W[1:]:
        M = A.reshape(A.shape[0], -1)
        np.testing.assert_allclose(M @ M.T, np.eye(A.shape[0]), atol=1e-10)
    # it is a combination of label-4 product states: its overlap with a label-4 image exceeds that with the others on average
    from oracle import pyoracle
    phi = pyoracle.features_single(pixels, True)
    o = pyoracle.SingleOracle(phi, labels, 4, W)
    f = np.array([o.output(i) for i in range(len(labels))])
    assert np.abs(f[labels == 4]).mean() >= np.abs(f[labels != 4]).mean()
    # deterministic in the seed
    out2 = str(tmp_path / "W4b")
    hostlib.build_initial_single(data, per_label, 4, 5, 7, True, out2)
    assert all(np.array_equal(a, b) for a, b in zip(W, hostlib.read_mps(out2)))


def test_per_label_launcher_plan_without_a_gpu(tmp_path):
    """BASELINE config 4 ("single.cc per-label MPS x10, one label per GPU"): `single` with `labels = all` is a launcher -- one child per
    label, at most one per GPU, the next label as soon as a device is free.  `dry_run = yes` prints the plan (no GPU, no data needed):
    ten labels over four devices from device 2 on, and a comma list over the default eight."""
    inp = tmp_path / "in"
    inp.write_text("input\n{\nlabels = all\nngpu = 4\ndevice = 2\ndry_run = yes\n}\n")
    run = subprocess.run([os.path.join(ROOT, "tnml_amd", "single"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "Per-label training of 10 labels on 4 GPUs" in run.stdout
    plan = re.findall(r"label (\d) -> device (\d), directory L(\d), writes L(\d)/W(\d)", run.stdout)
    assert [int(p[0]) for p in plan] == list(range(10)) and all(p[0] == p[2] == p[3] == p[4] for p in plan)
    assert [int(p[1]) for p in plan] == [2, 3, 4, 5, 2, 3, 4, 5, 2, 3]
    assert not any(os.path.isdir(tmp_path / ("L%d" % l)) for l in range(10))             # a dry run touches nothing
    inp.write_text("input\n{\nlabels = 3,7\ndry_run = yes\n}\n")
    run = subprocess.run([os.path.join(ROOT, "tnml_amd", "single"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert run.returncode == 0 and "Per-label training of 2 labels on 8 GPUs" in run.stdout
    assert re.findall(r"label (\d) -> device (\d)", run.stdout) == [("3", "0"), ("7", "1")]
    inp.write_text("input\n{\nlabels = 3,12\ndry_run = yes\n}\n")
    run = subprocess.run([os.path.join(ROOT, "tnml_amd", "single"), str(inp)], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert run.returncode != 0 and "not in 0..9" in run.stdout
