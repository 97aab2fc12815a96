mkdir -p gpurun_out/r5q
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r5q/pytest_all.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5q/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5q/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5q/smoke.txt
bash tools/final_r5.sh
