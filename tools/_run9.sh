python bench.py --per-layer > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_per_layer.txt
tail -c 300 gpurun_out/r04i_bench.json
