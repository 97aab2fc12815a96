export TNML_COMMIT=$(cat .commit_for_pmc 2>/dev/null || echo unknown)
bash tools/pmc_bench.sh > gpurun_out/pmc_r05.txt 2>&1
tail -40 gpurun_out/pmc_r05.txt | cut -c1-200
ls -la gpurun_out/pmc/
