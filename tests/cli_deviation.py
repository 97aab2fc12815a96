"""Deviation of the `fixedL` CLI driver's per-bond costs from the oracle on the tiny idx problem of
tests/test_gpu_parity.py::test_fixedl_cli_driver_end_to_end (test infrastructure: prints, asserts nothing)."""
import os, re, subprocess, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it calls the oracle)
sys.path.insert(0, root)
from oracle import pyoracle
from tnml_amd import hostlib, synth
tmp = tempfile.mkdtemp()
N, per_label = 16, 20
labels = synth.synthetic_labels(10 * per_label, seed=9, per_label=per_label)
pixels = synth.synthetic_images(N, labels, seed=9)
pixels = np.clip(pixels.astype(np.int32) * 3, 0, 255).astype(np.uint8)
data = os.path.join(tmp, "data")
synth.write_idx(data, pixels, labels)
inp = os.path.join(tmp, "input")
open(inp, "w").write("input\n{\ndatadir = %s\nNtrain = %d\nNbatch = 4\nNsweep = 1\ncutoff = 1E-10\nmaxm = 6\nminm = 3\n"
                     "ninitial = 3\nlambda = 1E-3\nNpass = 3\nseed = 5\nprecision = f64\n}\n" % (data, per_label))
run = subprocess.run([os.path.join(root, "tnml_amd", "fixedL"), inp], capture_output=True, text=True, cwd=tmp, timeout=300)
if run.returncode:
    print("fixedL failed", run.stderr[-1000:]); sys.exit(1)
costs = np.array([float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", run.stdout)])
newm = [int(x) for x in re.findall(r"New m=(\d+)", run.stdout)]
w0 = os.path.join(tmp, "W0ref")
hostlib.build_initial_w(data, per_label, 3, 5, w0)
px, lab, _ = hostlib.read_mnist(data, True, per_label)
o = pyoracle.Oracle(pyoracle.features_series(px), lab, hostlib.read_mps(w0))
o.init()
ro = o.mldmrg(1, 6, 3, 1e-10, 3, 1e-3, 1e-10)
ref_c = np.array([r["cost"] / len(lab) for r in ro]); ref_m = [r["newm"] for r in ro]
rel = np.abs(costs - ref_c) / np.abs(ref_c)
print("%s: max rel cost deviation %.2e (bonds 1-6: %.1e); m equal: %s" % (sys.argv[1] if len(sys.argv) > 1 else "", rel.max(), rel[:6].max(), newm == ref_m))
print("   per bond:", " ".join("%.0e" % r for r in rel))
print("   m gpu:", newm, "\n   m ref:", ref_m)
