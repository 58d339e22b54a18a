set -u
OLD="WDG_TUNE=tap_chunk_order=0,tap_class_order=0,wgrad_xcd=0,gather_xcd=0"
AB_STEPS=12 bash tools/ab_step.sh "new:" "oldorder:$OLD" "new:" "oldorder:$OLD" "new:" "oldorder:$OLD" > gpurun_out/r05p_ab_step_locality.txt 2>&1
cat gpurun_out/r05p_ab_step_locality.txt
