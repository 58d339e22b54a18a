O=gpurun_out/r04k_step16.txt
: > $O
python tools/perf_step16.py bf16 >> $O 2>&1
python tools/perf_step16.py bf16 patch_lstm_small=0 >> $O 2>&1
python tools/perf_step16.py bf16 patch_lstm_small=64 >> $O 2>&1
python tools/perf_step16.py bf16 patch_lstm_small=0 patch_lstm_small=64 >> $O 2>&1
python tools/perf_step16.py bf16 patch_lstm_small=128 >> $O 2>&1
python tools/perf_step16.py bf16 patch_lstm_small=0 patch_lstm_small=128 >> $O 2>&1
grep -v amdgpu.ids $O
