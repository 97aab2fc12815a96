#!/bin/bash
# the randomised differential runs of tests/debug/ with fresh seeds (HIP path against the oracle, bond by bond)
S=${1:-9001}
mkdir -p gpurun_out/fuzz
timeout 900 python tests/debug/fuzz_lockstep.py 150 $S > gpurun_out/fuzz/lockstep_$S.txt 2>&1; echo "lockstep rc=$?"; tail -1 gpurun_out/fuzz/lockstep_$S.txt; grep -A4 "^BAD\|raised" gpurun_out/fuzz/lockstep_$S.txt | head -40
FUZZ_STRICT=1 timeout 600 python tests/debug/fuzz_lockstep.py 100 $((S+1)) > gpurun_out/fuzz/strict_$S.txt 2>&1; echo "strict rc=$?"; tail -1 gpurun_out/fuzz/strict_$S.txt; grep -A4 "^BAD\|raised" gpurun_out/fuzz/strict_$S.txt | head -40
FUZZ_HUGE=1 timeout 900 python tests/debug/fuzz_lockstep.py 8 $((S+2)) > gpurun_out/fuzz/huge_$S.txt 2>&1; echo "huge rc=$?"; tail -1 gpurun_out/fuzz/huge_$S.txt; grep -A4 "^BAD\|raised" gpurun_out/fuzz/huge_$S.txt | head -40
timeout 900 python tests/debug/fuzz_single.py 60 $((S+3)) > gpurun_out/fuzz/single_$S.txt 2>&1; echo "single rc=$?"; tail -1 gpurun_out/fuzz/single_$S.txt; grep -A4 "^BAD\|raised" gpurun_out/fuzz/single_$S.txt | head -40
