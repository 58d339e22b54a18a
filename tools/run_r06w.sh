#!/bin/bash
for N in 16 32 64; do
  bash tools/pmc_infer.sh r06w$N bf16 $N trace > gpurun_out/r06w_trace$N.txt 2>&1
done
python3 tools/prof_infer_group.py bf16 10 graph 16 > gpurun_out/r06w_graph.txt 2>&1
python3 tools/prof_infer_group.py bf16 10 graph 32 >> gpurun_out/r06w_graph.txt 2>&1
python3 tools/prof_infer_group.py bf16 10 graph 64 >> gpurun_out/r06w_graph.txt 2>&1
python3 tools/prof_infer_group.py fp16 10 graph 64 >> gpurun_out/r06w_graph.txt 2>&1
