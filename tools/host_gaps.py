#!/usr/bin/env python3
"""For the largest GPU-idle gaps of a kernel trace: which HIP runtime calls the host was in during the gap (rocprofv3
--kernel-trace --hip-runtime-trace CSVs).   python tools/host_gaps.py kernel_trace.csv hip_api_trace.csv"""
import csv
import sys
from collections import Counter


def main():
    k = list(csv.DictReader(open(sys.argv[1])))
    a = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in k)
    api = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in a)
    t0, t1 = ev[0][0], max(e for _, e, _ in ev)
    cut = t1 - (t1 - t0) / 4
    evs = [x for x in ev if x[0] >= cut]
    cur, prev, gaps = evs[0][1], evs[0][2], []
    for s, e, n in evs[1:]:
        if s > cur:
            gaps.append((s - cur, cur, s, prev, n))
        if e > cur:
            cur, prev = e, n
    gaps.sort(reverse=True)
    tot = Counter()
    for g, gs, ge, before, after in gaps[:25]:
        inside = Counter()
        for s, e, f in api:
            if e < gs or s > ge:
                continue
            inside[f] += min(e, ge) - max(s, gs)
        top = ", ".join(f"{f} {v / 1e3:.0f}us" for f, v in inside.most_common(3))
        for f, v in inside.items():
            tot[f] += v
        print(f"gap {g / 1e3:8.1f} us  after {before[:36]:36s} before {after[:36]:36s} | host: {top}")
    print("host time inside the 25 largest gaps by call:", ", ".join(f"{f} {v / 1e6:.2f}ms" for f, v in tot.most_common(8)))
    # the longest host calls of the window overall
    longest = sorted(((e - s, f) for s, e, f in api if s >= cut), reverse=True)[:12]
    print("longest host calls:", ", ".join(f"{f} {d / 1e3:.0f}us" for d, f in longest))


if __name__ == "__main__":
    main()
