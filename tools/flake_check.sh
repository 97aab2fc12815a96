#!/bin/bash
# the -m gpu suite N times in a row on one box (flake hunt before the driver's round-end run); one line per run + the failures
N=${1:-3}
mkdir -p gpurun_out/flake
for i in $(seq 1 $N); do
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/flake/run$i.txt 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/flake/run$i.txt)"
  grep -E "^(FAILED|ERROR)" gpurun_out/flake/run$i.txt | head -10
done
