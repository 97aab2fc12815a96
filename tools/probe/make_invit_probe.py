"""writes tools/probe/_invit_var.inc: k_tridiag_invit of tnml_amd/csrc/eigh.hip as a template with cut points (VAR 1/2: return after
the factorisation, 3: one sweep instead of two, 4: no zero fill outside the block) for tools/probe/probe_invit.hip.
  python tools/probe/make_invit_probe.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Itnml_amd/csrc -Iinclude -DNOPROF \\
      tools/probe/probe_invit.hip -o tools/probe/probe_invit -lrocblas -lrocsolver"""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "tnml_amd", "csrc", "eigh.hip")).read()
body = src[src.index("#define IV_L 16"):src.index("int eigh_tridiag_eig(")]
def rep(a, b):
    global body
    assert a in body, a
    body = body.replace(a, b)
rep("__global__ __launch_bounds__(64) void k_tridiag_invit(TeigArgs T) {", "template <int VAR> __global__ __launch_bounds__(64) void k_tridiag_invit_v(TeigArgs T) {")
rep("#define IV_L 16", "#undef IV_L\n#define IV_L 16")
m1 = "    a[IX(hi - 1)] = rpiv(ak); b[IX(hi - 1)] = 0.; d2[IX(hi - 1)] = 0.;"
rep(m1, "    if (VAR == 1 || VAR == 2) { zc[lo] = ak; return; }\n" + m1)
rep("    for (int iter = 0; iter < 2; ++iter) {", "    for (int iter = 0; iter < (VAR == 3 ? 1 : 2); ++iter) {")
rep("    for (int k = 0; k < lo; ++k) zc[k] = 0.;\n    for (int k = hi; k < n; ++k) zc[k] = 0.;",
    "    if (VAR != 4) { for (int k = 0; k < lo; ++k) zc[k] = 0.;\n    for (int k = hi; k < n; ++k) zc[k] = 0.; }")
open(os.path.join(root, "tools", "probe", "_invit_var.inc"), "w").write(body)
