#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels in a device assembly listing:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/k.s wind-downscaling-gan_amd/csrc/<file>.hip
    python tools/kernel_regs.py /tmp/k.s [substring ...]
vgpr = arch VGPRs + AGPRs (unified file of 512 per SIMD lane: 128 -> 4 waves per SIMD, 168 -> 3, 256 -> 2)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"^void ", "", dn).split("(")[0]
    if want and not any(w in dn for w in want):
        continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\d+)" % k, body) or [None, "-"])[1]     # noqa: E731
    total = int(g("next_free_vgpr"))
    print(f"{dn[:64]:64s} vgpr {total:4d} (arch {g('accum_offset'):>3s}) waves/SIMD {min(8, 512 // max(total, 1)):d}  sgpr {g('next_free_sgpr'):>3s}  "
          f"scratch {g('private_segment_fixed_size'):>4s}  lds {g('group_segment_fixed_size'):>6s}")
