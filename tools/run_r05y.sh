set -u
AB_STEPS=10 bash tools/ab_step.sh "q4:" "q5:GPU_MAX_HW_QUEUES=5" "q6:GPU_MAX_HW_QUEUES=6" "q8:GPU_MAX_HW_QUEUES=8" "q5_wg:GPU_MAX_HW_QUEUES=5 WDG_TRIPLE_WGRAD=1" "q6_wg:GPU_MAX_HW_QUEUES=6 WDG_TRIPLE_WGRAD=1" "q4:" > gpurun_out/r05ba_ab.txt 2>&1; cut -c1-140 gpurun_out/r05ba_ab.txt
