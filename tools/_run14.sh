O=gpurun_out/r04n_h16.txt
: > $O
python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -3 >> $O
python tools/perf_step16.py bf16 2>&1 | grep -v amdgpu.ids >> $O
python tools/perf_step16.py fp16 2>&1 | grep -v amdgpu.ids >> $O
python tools/perf_patch.py 2>&1 | grep -v amdgpu.ids | tail -30 >> $O
python tools/run_configs.py gpurun_out/r04n_other_configs.json 2>&1 | grep -v amdgpu.ids | tail -5 >> $O
cat $O
