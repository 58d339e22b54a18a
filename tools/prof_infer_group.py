#!/usr/bin/env python3
"""The bench's config3_bf16_group16_T24 leg (one predict() group of 16 tiles x 24 h through the shipped generator at
inference precision, image + noise assembled on the device) with eager launches, for `rocprofv3 --kernel-trace --stats`:
    python tools/prof_infer_group.py [bf16|fp16] [reps] [graph|eager] [tiles]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    import downscaling.api as api
    network = api.get_network(allow_random_init=True, random_seed=5)
    gen = network.generator
    gen.inference_precision = prec
    gen.graph_inference = graph
    dev = gen.ops.device
    tiles = torch.randn(n, api.SEQUENCE_LENGTH, api.IMG_SIZE, api.IMG_SIZE, 3, device=dev)
    if n > 16:
        # several predict() groups of 16 in one forward pass, each with its own draw (api.predict_array's default is 2)
        from downscaling.data.data_generator import LazyGroupNoise
        ng = network.noise_generator
        fn = lambda: gen([tiles, LazyGroupNoise(ng, n // 16, 16, ng.noise_shape, api.NOISE_CHANNELS, ng.std)])   # noqa: E731
    else:
        fn = lambda: gen([tiles, network.noise_generator.lazy(bs=n, channels=api.NOISE_CHANNELS)])   # noqa: E731
    for _ in range(4):          # (graph mode: eager, eager, capture, first replay)
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{prec} graph={graph}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per {n}-tile group")


if __name__ == "__main__":
    main()
