set -u
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gates_x or layer_sequences" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "discriminator or default_widths or graphs or test_train_step" 2>&1 | tail -3
T="-- --size 96 --timesteps 24 --batch 8"
AB_STEPS=8 bash tools/ab_step.sh "t24new:$T" "t24old:WDG_DHIGH_DIRECT=0 $T" "t24new:$T" "t24old:WDG_DHIGH_DIRECT=0 $T" > gpurun_out/r05av_ab.txt 2>&1; cut -c1-60 gpurun_out/r05av_ab.txt
