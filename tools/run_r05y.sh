set -u
AB_STEPS=12 bash tools/ab_step.sh "both:" "g:WDG_WGRAD_STREAM_ONLY=g" "d:WDG_WGRAD_STREAM_ONLY=d" "both:" "g:WDG_WGRAD_STREAM_ONLY=g" "d:WDG_WGRAD_STREAM_ONLY=d" > gpurun_out/r05y2_ab.txt 2>&1; cut -c1-20 gpurun_out/r05y2_ab.txt
python bench.py > gpurun_out/r05y_bench.txt 2>&1; tail -1 gpurun_out/r05y_bench.txt | cut -c1-1500
