set -u
WDG_XSTEP=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
T="-- --size 96 --timesteps 24 --batch 8"
AB_STEPS=10 bash tools/ab_step.sh "new:" "xstep:WDG_XSTEP=1" "new:" "xstep:WDG_XSTEP=1" "t24new:$T" "t24xstep:WDG_XSTEP=1 $T" "t24new:$T" "t24xstep:WDG_XSTEP=1 $T" > gpurun_out/r05ah_ab.txt 2>&1; cut -c1-60 gpurun_out/r05ah_ab.txt
