mkdir -p gpurun_out/r6n
for n in 2100 7500 60000; do python tools/dev_grad.py $n 20 > gpurun_out/r6n/grad$n.txt 2>&1; echo "== $n"; grep -v "amdgpu.ids" gpurun_out/r6n/grad$n.txt | tail -3; done
python -m pytest tests/test_configs_at_shape.py -m gpu -x -q -k "gradient_quad" 2>&1 | tail -3
