mkdir -p gpurun_out/r6d
export TNML_T_TIMEOUT=10
(TNML_T_DEPTH=2 timeout 200 python tools/oneshot_processes_m120.py 2 1536 8 3) > gpurun_out/r6d/os2_d2.txt 2>&1
(TNML_T_DEPTH=2 timeout 300 python tools/oneshot_processes_m120.py 3 1536 8 3) > gpurun_out/r6d/os3_d2.txt 2>&1
(TNML_T_DEPTH=1 TNML_SPEC_SPLIT=0 timeout 100 python tools/oneshot_processes_m120.py 2 1536 8 1) > gpurun_out/r6d/os2_d1_nospec.txt 2>&1
for f in os2_d2 os3_d2 os2_d1_nospec; do echo "== $f"; grep "^run\|runs clean" gpurun_out/r6d/$f.txt | cut -c1-300; done
TNML_DEV_ABL=1 python tools/dev_grad.py 60000 20 > gpurun_out/r6d/grad60000.txt 2>&1
TNML_DEV_ABL=1 python tools/dev_grad.py 7500 20 > gpurun_out/r6d/grad7500.txt 2>&1
for f in grad7500 grad60000; do echo "== $f"; grep -v "amdgpu.ids" gpurun_out/r6d/$f.txt | tail -6; done
