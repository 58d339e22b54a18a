#!/usr/bin/env python3
"""Summary of a rocprofv3 kernel trace (CSV with Start_Timestamp / End_Timestamp / Queue_Id per dispatch): wall time of the
last traced step (between the two largest idle gaps = the bench's step boundaries are not marked, so the step is taken as the
last 1/N of the busy span), GPU-busy time (union of the dispatch intervals), time with >= 2 / >= 3 kernels resident, the
largest idle gaps and what ran around them.   python tools/timeline_summary.py trace.csv [n_steps_in_trace=4]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, e, r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
    ev.sort()
    t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
    # keep the last step's share of the span
    cut = t1 - (t1 - t0) / nsteps
    evs = [x for x in ev if x[0] >= cut]
    span = (max(e for _, e, _, _ in evs) - evs[0][0]) / 1e6
    pts = []
    for s, e, _, _ in evs:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    busy = [0.0, 0.0, 0.0, 0.0]
    depth, last = 0, pts[0][0]
    gaps = []
    for t, d in pts:
        if depth >= 1:
            busy[min(depth, 3)] += t - last
        elif t > last:
            gaps.append((t - last, last))
        depth += d
        last = t
    tot_busy = sum(busy[1:]) / 1e6
    print(f"dispatches in the window: {len(evs)}; window {span:.2f} ms; GPU busy (>= 1 kernel resident) {tot_busy:.2f} ms; "
          f"exactly 1: {busy[1] / 1e6:.2f} ms, exactly 2: {busy[2] / 1e6:.2f} ms, >= 3: {busy[3] / 1e6:.2f} ms")
    ksum = sum(e - s for s, e, _, _ in evs) / 1e6
    print(f"sum of kernel durations {ksum:.2f} ms; idle inside the window {span - tot_busy:.2f} ms in {len(gaps)} gaps")
    queues = {}
    for s, e, _, q in evs:
        queues[q] = queues.get(q, 0) + (e - s)
    print("per queue busy ms:", {q: round(v / 1e6, 2) for q, v in sorted(queues.items())})
    # which kernels run ALONE (no other kernel resident), and for how long: the serial sections of the schedule — a small kernel in
    # this list is on the critical path (removing it shortens the step), one that never runs alone is hidden
    bounds = sorted({t for s, e, _, _ in evs for t in (s, e)})
    import bisect
    alone = {}
    active = []
    evs_by_start = sorted(evs)
    idx = 0
    import heapq
    heap = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        while idx < len(evs_by_start) and evs_by_start[idx][0] <= a:
            heapq.heappush(heap, (evs_by_start[idx][1], evs_by_start[idx][2]))
            idx += 1
        while heap and heap[0][0] <= a:
            heapq.heappop(heap)
        if len(heap) == 1:
            alone[heap[0][1]] = alone.get(heap[0][1], 0) + (b - a)
    tot_alone = sum(alone.values()) / 1e6
    print(f"time with exactly one kernel resident, by kernel (total {tot_alone:.2f} ms):")
    for name, v in sorted(alone.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {v / 1e6:7.3f} ms  {name}")
    gaps.sort(reverse=True)
    for g, at in gaps[:8]:
        before = [n for s, e, n, _ in evs if e <= at + 1 and e >= at - 1]
        after = [n for s, e, n, _ in evs if s >= at + g - 1 and s <= at + g + 1]
        print(f"  gap {g / 1e3:8.1f} us  after {before[:1]}  before {after[:1]}")


if __name__ == "__main__":
    main()
