"""Prints the device timeline of the last calls in a rocprofv3 kernel + memory-copy trace: start (us), duration, gap to the previous event, name."""
import csv
import sys

d, prefix, last = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40
ev = []
for r in csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")):
    n = r["Kernel_Name"].replace("pairhmm_hip::(anonymous namespace)::", "").replace("pairhmm_hip::", "").split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
try:
    for r in csv.DictReader(open(f"{d}/{prefix}_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]))
except FileNotFoundError:
    pass
ev.sort()
base = ev[-last][0]
prev = None
for s, e, n in ev[-last:]:
    print(f"{(s - base) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {((s - prev) / 1e3 if prev else 0):7.1f}  {n[:70]}")
    prev = e
