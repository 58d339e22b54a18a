set -u
python -m pytest tests -x -q -m gpu > gpurun_out/r05ad_tests.txt 2>&1; tail -3 gpurun_out/r05ad_tests.txt
T="-- --size 96 --timesteps 24 --batch 8"
AB_STEPS=10 bash tools/ab_step.sh "new:" "gs:WDG_TRIPLE_WGRAD=gs" "gs_real:WDG_TRIPLE_WGRAD=gs_real" "new:" "gs:WDG_TRIPLE_WGRAD=gs" "gs_real:WDG_TRIPLE_WGRAD=gs_real" "t24new:$T" "t24gs:WDG_TRIPLE_WGRAD=gs $T" "t24gs_real:WDG_TRIPLE_WGRAD=gs_real $T" > gpurun_out/r05ae_ab.txt 2>&1; cut -c1-60 gpurun_out/r05ae_ab.txt
