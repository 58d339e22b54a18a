set -u
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "upsample or upconv" 2>&1 | tail -3
bash tools/pmc_ops.sh r05m upconv_fwd "xcd:gather_xcd=1" "lin:gather_xcd=0"
