import os, re, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sklearn.datasets import load_digits
from oracle import pyoracle
from tnml_amd import hostlib, synth
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
d = load_digits()
px = np.clip(np.rint(d.images.reshape(-1, 64) * (255.0 / 16.0)), 0, 255).astype(np.uint8)
lab = d.target.astype(np.int32)
per_label = 150
train_idx = np.sort(np.concatenate([np.flatnonzero(lab == l)[:per_label] for l in range(10)]))
data = tmp + "/data"
synth.write_idx(data, px[train_idx], lab[train_idx], side=8)
keys = "datadir = %s\nfeature_scale = 255\n" % data
open(tmp + "/input", "w").write("input\n{\n%sNtrain = %d\nNbatch = 10\nNsweep = 1\ncutoff = 1E-10\nmaxm = 10\nminm = 5\nninitial = 5\nlambda = 1E-3\nNpass = 4\nseed = 3\n}\n" % (keys, per_label))
run = subprocess.run([root + "/tnml_amd/fixedL", tmp + "/input"], capture_output=True, text=True, cwd=tmp)
log = run.stdout
print("\n".join(log.split("\n")[:60]))
def feats(p):
    g = p.astype(np.float64) / 255.0
    return np.stack([np.ones_like(g), 255.0 * ((g / 255.0) / 4.0)], axis=-1)
w0 = tmp + "/W0ref"
hostlib.build_initial_w(data, per_label, 5, 3, w0, feature_scale=255.0)
trp, trl, _ = hostlib.read_mnist(data, True, per_label)
print("train images", trp.shape, "identical to source:", np.array_equal(trp, px[train_idx]))
o = pyoracle.Oracle(feats(trp), trl, hostlib.read_mps(w0), nthread=1)
o.init()
B = o.bond_tensor(1)
print("oracle before cost", o.quadcost(B, 1e-3)[0] / len(trl))
Bo, tr = o.cgrad(B, 4, 1e-3, 1e-10)
print("oracle cg", [c / len(trl) for c in tr["cost"]], tr["rnorm"], tr["alpha"])
ro = o.mldmrg(1, 10, 5, 1e-10, 4, 1e-3, 1e-10, max_bonds=3)
for r in ro: print("oracle bond", r["bond"], r["cost"] / len(trl), r["newm"], r["truncerr"])
