set -u
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/pmc_ops.sh r05n g5_fwd,g5_fwd_bn,g4_fwd,g4_dgrad,g5_wgrad,g4_wgrad "chunk:tap_chunk_order=1" "tapmajor:tap_chunk_order=0"
