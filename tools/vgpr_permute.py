#!/usr/bin/env python3
"""Renumbers the vector registers of one kernel in gfx950 assembly text.

On gfx950 a VALU instruction that reads three registers of one VGPR bank (register number mod 4) issues at
half rate (tools/gen_bank_microbench.py).  The compiler's allocator does not look at banks.  A renumbering of
a kernel's VGPRs, applied to every instruction of the kernel alike, leaves the program unchanged and moves
operands between banks; this tool applies one (random from a seed, or read from a file) and reports the bank
conflicts of the kernel's hottest loop before and after.

Registers that appear together in a tuple v[a:b] stay together, in order, and keep the parity of their first
register (64-bit operands must start on an even register).  v0 holds the work-item id at entry and stays.

usage: vgpr_permute.py in.s out.s --kernel SUBSTRING [--seed N | --map FILE] [--loop LABEL] [--save-map FILE]
"""
import argparse
import json
import random
import re
import sys

REG = re.compile(r"\bv(\d+)\b")
TUP = re.compile(r"\bv\[(\d+):(\d+)\]")


def split_comment(line):
    k = line.find(";")
    return (line, "") if k < 0 else (line[:k], line[k:])


def kernel_span(lines, needle):
    """[begin, end) line indices of the body of the first function whose label contains needle."""
    begin = None
    for i, line in enumerate(lines):
        if begin is None and needle in line and re.match(r"^[A-Za-z_.$][\w.$]*:", line):
            begin = i
        elif begin is not None and line.startswith(".Lfunc_end"):
            return begin, i
    raise SystemExit(f"kernel {needle!r} not found")


def units_of(body):
    """Maximal runs of registers tied together by tuples, as (start, size, in_tuple) sorted by start."""
    used, tied = set(), {}
    for line in body:
        code, _ = split_comment(line)
        for a, b in TUP.findall(code):
            a, b = int(a), int(b)
            used.update(range(a, b + 1))
            for r in range(a, b):
                tied[r] = True  # r and r + 1 stay adjacent
        for r in REG.findall(code):
            used.add(int(r))
    units, r, top = [], 0, max(used)
    while r <= top:
        if r not in used:
            r += 1
            continue
        s = r
        while tied.get(r):
            r += 1
        units.append((s, r - s + 1))
        r += 1
    return units, top


def random_map(units, top, limit, rng):
    """A random placement of the units in [0, limit], packed in a random order: {old register: new register}.
    Pairs of single registers dropped between tuples shift what follows by two, which is what moves a
    four-register tuple to the other pair of banks."""
    for _ in range(1000):
        fixed = [u for u in units if u[0] == 0]
        tuples = [u for u in units if u[0] != 0 and u[1] > 1]
        singles = [u for u in units if u[0] != 0 and u[1] == 1]
        rng.shuffle(tuples)
        rng.shuffle(singles)
        mapping = {}
        cursor = 0
        for s, n in fixed:
            for k in range(n):
                mapping[s + k] = s + k
            cursor = max(cursor, s + n)
        for s, n in tuples:
            if cursor % 2 != s % 2:
                if singles:
                    mapping[singles.pop()[0]] = cursor
                cursor += 1
            elif len(singles) >= 2 and rng.random() < 0.3:
                mapping[singles.pop()[0]] = cursor
                mapping[singles.pop()[0]] = cursor + 1
                cursor += 2
            for k in range(n):
                mapping[s + k] = cursor + k
            cursor += n
        for s, _ in singles:
            mapping[s] = cursor
            cursor += 1
        if cursor - 1 <= limit:
            return mapping
    raise SystemExit("no placement found")


def apply_map(body, mapping):
    out = []
    for line in body:
        code, comment = split_comment(line)
        code = TUP.sub(lambda m: f"v[{mapping[int(m.group(1))]}:{mapping[int(m.group(1))] + int(m.group(2)) - int(m.group(1))}]", code)
        code = REG.sub(lambda m: f"v{mapping[int(m.group(1))]}", code)
        out.append(code + comment)
    return out


def hottest_loop(body, label):
    """Lines of the loop that starts at `label` (up to the branch back to it); without a label, the
    innermost loop with the most VALU instructions."""
    starts = {}
    best = (0, 0, 0)
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            starts[m.group(1)] = i
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", line)
        if m and m.group(1) in starts and (label is None or m.group(1) == label):
            b = starts[m.group(1)]
            inner = body[b + 1:i]
            if any(re.match(r"^\.LBB", x) for x in inner):
                continue
            valu = sum(1 for x in inner if re.match(r"\s+v_", x))
            if valu > best[0]:
                best = (valu, b, i)
    return body[best[1]:best[2]]


def conflicts(loop):
    """(instructions reading three registers of one bank, two-source instructions whose sources share a bank)."""
    three = two = 0
    for line in loop:
        code, _ = split_comment(line)
        m = re.match(r"\s+(v_\w+)\s+(.*)", code)
        if not m or "dpp" in m.group(1) or m.group(1).startswith(("v_mov", "v_pk")):
            continue
        args = [a.strip() for a in m.group(2).split(",")]
        reads = args[1:] + ([args[0]] if "fmac" in m.group(1) else [])
        regs = sorted({int(a[1:]) for a in reads if re.fullmatch(r"v\d+", a)})
        banks = [r % 4 for r in regs]
        if any(banks.count(b) >= 3 for b in range(4)):
            three += 1
        srcs = [int(a[1:]) for a in args[1:3] if re.fullmatch(r"v\d+", a)]
        if len(srcs) == 2 and srcs[0] != srcs[1] and srcs[0] % 4 == srcs[1] % 4:
            two += 1
    return three, two


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--kernel", required=True, help="substring of the kernel's mangled name")
    ap.add_argument("--seed", type=int)
    ap.add_argument("--map", help="JSON {old: new} to apply instead of a random renumbering")
    ap.add_argument("--save-map")
    ap.add_argument("--loop", help="label of the loop to report conflicts for (default: the largest inner loop)")
    args = ap.parse_args()

    lines = open(args.src).read().split("\n")
    begin, end = kernel_span(lines, args.kernel)
    body = lines[begin:end]
    units, top = units_of(body)
    limit = (top // 8) * 8 + 7  # the allocation granule the kernel already pays for
    if args.map:
        mapping = {int(k): int(v) for k, v in json.load(open(args.map)).items()}
    elif args.seed is not None:
        mapping = random_map(units, top, limit, random.Random(args.seed))
    else:
        mapping = {r: r for s, n in units for r in range(s, s + n)}
    assert len(set(mapping.values())) == len(mapping) and max(mapping.values()) <= limit
    new_body = apply_map(body, mapping)
    before, after = conflicts(hottest_loop(body, args.loop)), conflicts(hottest_loop(new_body, args.loop))
    new_top = max(mapping.values())
    out = lines[:begin] + new_body + lines[end:]
    if new_top != top:  # the kernel descriptor and the metadata state the register count
        text = "\n".join(out)
        name = re.match(r"^([\w.$]+):", body[0]).group(1)
        k0 = text.index(f".amdhsa_kernel {name}")
        k1 = text.index(".end_amdhsa_kernel", k0)
        desc = re.sub(r"(\.amdhsa_next_free_vgpr\s+)\d+", lambda m: m.group(1) + str(new_top + 1), text[k0:k1])
        desc = re.sub(r"(\.amdhsa_accum_offset\s+)\d+", lambda m: m.group(1) + str((new_top // 4 + 1) * 4), desc)
        text = text[:k0] + desc + text[k1:]
        m0 = re.search(r"\.name:\s+" + re.escape(name) + r"\n", text).start()
        m1 = text.index(".vgpr_count:", m0)
        text = text[:m1] + re.sub(r"\.vgpr_count:\s+\d+", f".vgpr_count:     {new_top + 1}", text[m1:], count=1)
        out = text.split("\n")
    open(args.dst, "w").write("\n".join(out))
    if args.save_map:
        json.dump({str(k): v for k, v in sorted(mapping.items())}, open(args.save_map, "w"))
    print(json.dumps({"units": len(units), "registers": len(mapping), "top_before": top, "top_after": new_top,
                      "three_in_one_bank_before": before[0], "two_sources_one_bank_before": before[1],
                      "three_in_one_bank_after": after[0], "two_sources_one_bank_after": after[1]}))


if __name__ == "__main__":
    main()
