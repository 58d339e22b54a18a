O=gpurun_out/r04o_h16.txt
: > $O
python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -2 >> $O
python tools/perf_step16.py bf16 2>&1 | grep -v amdgpu.ids >> $O
python tools/perf_step16.py fp16 2>&1 | grep -v amdgpu.ids >> $O
python tools/_probe_chain.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
