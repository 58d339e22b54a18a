set -u
AB_STEPS=20 bash tools/ab_step.sh "new:" "full:WDG_LSTM_LIVE_GATES=0" "new:" "full:WDG_LSTM_LIVE_GATES=0" "new:" "full:WDG_LSTM_LIVE_GATES=0" > gpurun_out/r05al_ab.txt 2>&1; cut -c1-60 gpurun_out/r05al_ab.txt
