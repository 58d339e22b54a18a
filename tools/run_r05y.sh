set -u
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gates_x or layer_sequences" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "discriminator or default_widths or graphs" 2>&1 | tail -3
T="-- --size 96 --timesteps 24 --batch 8"
AB_STEPS=8 bash tools/ab_step.sh "t24new:$T" "t24old:$T --tune lstm2_thin=0" "t24new:$T" "t24old:$T --tune lstm2_thin=0" > gpurun_out/r05as_ab.txt 2>&1; cut -c1-60 gpurun_out/r05as_ab.txt
