#!/bin/bash
# compile-time variants of k_grad_quad (staging slot of a row group, issue priority while staging) on the box: rebuild kernels_grad.o with
# -D..., relink, time with tools/dev_grad.py at m = 120 (60 000 and 7 500 images); the default build is restored at the end
cd $GRAFT_REPO_ROOT/tnml_amd/csrc
run() {
  echo "=== $1"
  rm -f kernels_grad.o ../libtnml.so
  t0=$(date +%s)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off "$@" -c kernels_grad.hip -o kernels_grad.o 2>&1 | grep -E "error" | head -3
  make > /dev/null 2>&1 || echo LINK FAILED
  echo "    built in $(( $(date +%s) - t0 )) s"
  (cd $GRAFT_REPO_ROOT && python tools/dev_grad.py 60000 30 120 2>&1 | grep "k_grad_quad\|max |G" ; python tools/dev_grad.py 7500 30 120 2>&1 | grep "k_grad_quad")
}
run
run -DGQ_PRIO=0
run -DGQ_PRIO=1
run "-DGQ_SLOT(rgp)=(2*(rgp)+1)"
run "-DGQ_SLOT(rgp)=(rgp)"
run "-DGQ_SLOT(rgp)=(6-2*(rgp))"
run "-DGQ_SLOT(rgp)=((rgp)+2)"
run "-DGQ_SLOT(rgp)=(0)"
run
