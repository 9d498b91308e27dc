"""Summarises the calls in a rocprofv3 kernel + memory-copy trace: events separated by more than GAP_US of device
idle time start a new call; per call: span from first to last event, summed busy time (union of the events), the
largest idle gaps inside the call.  usage: trace_gaps.py DIR PREFIX [GAP_US]"""
import csv
import re
import sys

d, prefix = sys.argv[1], sys.argv[2]
call_gap = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 400e3
ev = []
for r in csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.match(r"[\w:]+", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")).group(0).split("::")[-1]))
try:
    for r in csv.DictReader(open(f"{d}/{prefix}_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]))
except FileNotFoundError:
    pass
ev.sort()
calls, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > call_gap:
        calls.append(cur)
        cur = []
    cur.append(e)
    end = e[1] if end is None else max(end, e[1])
calls.append(cur)
for c in calls[-4:]:
    first, last = c[0][0], max(e[1] for e in c)
    busy, gaps, reach = 0, [], c[0][0]
    for s, e, n in c:
        if s > reach:
            gaps.append(((s - reach) / 1e3, (reach - first) / 1e3, n))
            reach = s
        if e > reach:
            busy += e - reach
            reach = e
    gaps.sort(reverse=True)
    print(f"call of {len(c)} events: span {(last - first) / 1e3:.0f} us, busy {busy / 1e3:.0f} us, first event {c[0][2]}, last {c[-1][2]}")
    for g, at, n in gaps[:6]:
        print(f"    idle {g:7.1f} us at +{at:8.1f} us before {n[:50]}")
