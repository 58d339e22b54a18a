set -u
AB_STEPS=12 bash tools/ab_step.sh "new:" "ws2:WDG_WGRAD_STREAM=2" "ws1:WDG_WGRAD_STREAM=1" "new:" "ws2:WDG_WGRAD_STREAM=2" "ws1:WDG_WGRAD_STREAM=1" > gpurun_out/r05x_ab.txt 2>&1; cat gpurun_out/r05x_ab.txt
