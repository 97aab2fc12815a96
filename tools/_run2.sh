mkdir -p gpurun_out/r6b
export TNML_T_TIMEOUT=15
(TNML_T_DEPTH=1 timeout 120 python tools/oneshot_processes_m120.py 2 1536 8 1) > gpurun_out/r6b/os2_d1.txt 2>&1
(TNML_T_DEPTH=2 timeout 120 python tools/oneshot_processes_m120.py 2 1536 8 1) > gpurun_out/r6b/os2_d2.txt 2>&1
(TNML_T_DEPTH=1 timeout 150 python tools/oneshot_processes_m120.py 3 1536 8 1) > gpurun_out/r6b/os3_d1.txt 2>&1
(TNML_T_DEPTH=1 TNML_SPEC_SPLIT=0 timeout 120 python tools/oneshot_processes_m120.py 2 1536 8 1) > gpurun_out/r6b/os2_d1_nospec.txt 2>&1
for f in os2_d1 os2_d2 os3_d1 os2_d1_nospec; do echo "== $f"; grep -v "amdgpu.ids" gpurun_out/r6b/$f.txt | cut -c1-300 | tail -25; done
