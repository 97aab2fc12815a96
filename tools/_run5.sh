mkdir -p gpurun_out/r6l
for n in 7500 60000; do TNML_DEV_ABL=1 python tools/dev_grad.py $n 20 > gpurun_out/r6l/grad$n.txt 2>&1; echo "== $n"; grep -v "amdgpu.ids" gpurun_out/r6l/grad$n.txt | tail -7; done
