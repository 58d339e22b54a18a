set -u
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/pmc_ops.sh r05o g4_dgrad,g4_wgrad,g2_wgrad,g0_wgrad,d2_wgrad "x64:wgrad_xcd=64" "x32:wgrad_xcd=32"
