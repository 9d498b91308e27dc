"""From a rocprofv3 kernel trace of many small concurrent passes: per kernel name the launch count and mean duration,
and over the busy window the fraction of time with 0, 1, 2, 3+ forward kernels running at once.
usage: trace_overlap.py DIR PREFIX [NAME_SUBSTRING_OF_THE_FORWARD_KERNEL]"""
import collections
import csv
import re
import sys

d, prefix = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "fwd_"
rows = list(csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")))
by = collections.defaultdict(list)
edges = []
for r in rows:
    n = re.match(r"[\w:]+", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")).group(0).split("::")[-1]
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
# a short window of the timeline, one line per event, with the queue it ran on
if len(sys.argv) > 4:
    ev = []
    for r in rows:
        n = re.match(r"[\w:]+", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")).group(0).split("::")[-1]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
    try:
        for r in csv.DictReader(open(f"{d}/{prefix}_memory_copy_trace.csv")):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"].replace("MEMORY_COPY_", ""), "-"))
    except FileNotFoundError:
        pass
    ev.sort()
    mid = len(ev) * 3 // 4
    base = ev[mid][0]
    for s, e, n, q in ev[mid:mid + int(sys.argv[4])]:
        print(f"{(s - base) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  q{q:>3s}  {n}")
