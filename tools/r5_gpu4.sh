mkdir -p gpurun_out/r5d
for sg in 0 1; do
  TNML_SMALL_GEMM=$sg timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "hard_spectra" 2>&1 | tail -8 > gpurun_out/r5d/hard_sg$sg.txt
  echo "small_gemm=$sg"; cat gpurun_out/r5d/hard_sg$sg.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r5d/counters.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/r5d/counters.txt
