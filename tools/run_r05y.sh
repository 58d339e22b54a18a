set -u
true
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "discriminator or default_widths or graphs or test_train_step" 2>&1 | tail -3
T="-- --size 96 --timesteps 24 --batch 8"
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
