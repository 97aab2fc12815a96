#!/bin/bash
# compile-time variants of k_grad_quad on the box: rebuild kernels_grad.o with -D..., relink, time with tools/dev_grad.py at the bond
# dimensions given (default 120; 60 000 and 7 500 images); usage: tune_grad_quad.sh "<m list>" <variant flags, one argument per variant ("" = shipped)>
cd $GRAFT_REPO_ROOT/tnml_amd/csrc
MS=${1:-120}; shift
run() {
  echo "=== ${1:-shipped}"
  rm -f kernels_grad.o ../libtnml.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $1 -c kernels_grad.hip -o kernels_grad.o 2>&1 | grep -E "error" | head -3
  make > /dev/null 2>&1 || echo LINK FAILED
  for m in $MS; do (cd $GRAFT_REPO_ROOT && python tools/dev_grad.py 60000 30 $m 2>&1 | grep "k_grad_quad\|max |G" ; python tools/dev_grad.py 7500 30 $m 2>&1 | grep "k_grad_quad"); done
}
for v in "$@"; do run "$v"; done
