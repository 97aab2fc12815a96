import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle
from tnml_amd import hostlib, synth
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
N, per_label = 16, 20
labels = synth.synthetic_labels(10 * per_label, seed=9, per_label=per_label)
pixels = synth.synthetic_images(N, labels, seed=9)
pixels = np.clip(pixels.astype(np.int32) * 3, 0, 255).astype(np.uint8)
data = tmp + "/data"
synth.write_idx(data, pixels, labels)
open(tmp + "/input", "w").write("input\n{\ndatadir = %s\nNtrain = %d\nNbatch = 4\nNsweep = 1\ncutoff = 1E-10\nmaxm = 6\nminm = 3\nninitial = 3\nlambda = 1E-3\nNpass = 3\nseed = 5\nprecision = %s\n}\n" % (data, per_label, sys.argv[1] if len(sys.argv) > 1 else "strict"))
run = subprocess.run([root + "/tnml_amd/fixedL", tmp + "/input"], capture_output=True, text=True, cwd=tmp)
log = run.stdout
costs = [float(x) for x in re.findall(r"--> After SVD, Cost = ([0-9.eE+-]+)", log)]
newm = [int(x) for x in re.findall(r"New m=(\d+)", log)]
te = re.findall(r"SVD trunc err = ([0-9.eE+-]+)", log)
w0 = tmp + "/W0ref"
hostlib.build_initial_w(data, per_label, 3, 5, w0)
px, lab, _ = hostlib.read_mnist(data, True, per_label)
o = pyoracle.Oracle(pyoracle.features_series(px), lab, hostlib.read_mps(w0)); o.init()
ro = o.mldmrg(1, 6, 3, 1e-10, 3, 1e-3, 1e-10)
for k, r in enumerate(ro):
    print(k, r["bond"], r["half"], "m", newm[k], r["newm"], "te", te[k], "%.2e" % r["truncerr"], "cost", costs[k], r["cost"] / len(lab), "cg", ["%.8f" % (c / len(lab)) for c in r["cg"]["cost"][:2]])
print(log[-1500:] if len(sys.argv) > 2 else "")
