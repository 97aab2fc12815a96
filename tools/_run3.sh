mkdir -p gpurun_out/r6c
python tools/dev_grad.py 2100 5 > gpurun_out/r6c/grad2100.txt 2>&1
python tools/dev_grad.py 7500 20 > gpurun_out/r6c/grad7500.txt 2>&1
python tools/dev_grad.py 60000 20 > gpurun_out/r6c/grad60000.txt 2>&1
export TNML_T_TIMEOUT=10 TNML_IPC_TRACE=1
(TNML_T_DEPTH=1 TNML_SPEC_SPLIT=0 timeout 100 python tools/oneshot_processes_m120.py 2 1536 5 1) > gpurun_out/r6c/os2_d1_nospec.txt 2>&1
(TNML_T_DEPTH=2 timeout 100 python tools/oneshot_processes_m120.py 2 1536 5 1) > gpurun_out/r6c/os2_d2.txt 2>&1
for f in grad2100 grad7500 grad60000; do echo "== $f"; grep -v "amdgpu.ids" gpurun_out/r6c/$f.txt | tail -6; done
