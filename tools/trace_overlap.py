"""From a rocprofv3 kernel trace of many small concurrent passes: per kernel name the launch count and mean duration,
and over the busy window the fraction of time with 0, 1, 2, 3+ forward kernels running at once.
usage: trace_overlap.py DIR PREFIX [NAME_SUBSTRING_OF_THE_FORWARD_KERNEL]"""
import collections
import csv
import sys

d, prefix = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "fwd_"
rows = list(csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")))
by = collections.defaultdict(list)
edges = []
for r in rows:
    n = r["Kernel_Name"].split("::")[-1].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    by[n].append(e - s)
    if key in n:
        edges.append((s, 1))
        edges.append((e, -1))
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:60]:60s} n={len(v):6d} mean {sum(v) / len(v) / 1e3:8.1f} us  total {sum(v) / 1e6:8.2f} ms")
edges.sort()
# drop the warm-up: the window between the 10th and the 90th percentile of the forward launches
lo, hi = edges[len(edges) // 10][0], edges[len(edges) * 9 // 10][0]
level, prev, hist = 0, None, collections.Counter()
for t, step in edges:
    if prev is not None and lo <= prev and t <= hi:
        hist[min(level, 4)] += t - prev
    level += step
    prev = t
tot = sum(hist.values())
print("forward kernels running at once, share of the window:", {k: round(v / tot, 3) for k, v in sorted(hist.items())}, f"window {tot / 1e6:.1f} ms")
