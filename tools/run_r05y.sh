set -u
T="-- --size 96 --timesteps 24 --batch 8"
AB_STEPS=8 bash tools/ab_step.sh "t24:$T" "t24nograph:WDG_CHAIN_GRAPHS=0 $T" "t24:$T" "t24nograph:WDG_CHAIN_GRAPHS=0 $T" "t24twin_nograph:WDG_CHAIN_GRAPHS=0 WDG_OVERLAP_DISC=1 $T" > gpurun_out/r05ap_ab.txt 2>&1; cut -c1-60 gpurun_out/r05ap_ab.txt
