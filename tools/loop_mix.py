"""Instruction mix of the hottest loop of one kernel in a device assembly listing.

usage: loop_mix.py <file.s> <substring of the mangled kernel name>
The hottest loop is taken to be the basic block with the most fma instructions.
"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(_Z\S*%s\S*):.*?^\.Lfunc_end\d+:" % re.escape(key), text, re.S | re.M)
if not m:
    sys.exit("kernel not found")
blocks, cur, name = [], [], "entry"
for line in m.group(0).split("\n"):
    if re.match(r"^\.LBB\d+_\d+:", line):
        blocks.append((name, cur))
        name, cur = line.strip(), []
    elif line.startswith("\t") and not line.strip().startswith((".", ";")):
        cur.append(line.strip())
blocks.append((name, cur))
name, best = max(blocks, key=lambda b: sum("fma" in x.split()[0] for x in b[1]))
ops = collections.Counter(x.split()[0] for x in best)
groups = collections.Counter()
for k, v in ops.items():
    g = "valu" if k.startswith("v_") else "lds" if k.startswith("ds_") else "salu" if k.startswith("s_") else "other"
    groups[g] += v
print(m.group(1))
print(name, len(best), "instructions", dict(groups))
for k, v in sorted(ops.items(), key=lambda x: -x[1]):
    print(f"  {v:4d} {k}")
if len(sys.argv) > 3:
    print("\n".join(best))
