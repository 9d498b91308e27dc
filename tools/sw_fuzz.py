"""Seeded fuzz of the Smith-Waterman aligner against the oracle's restatement: random lengths, mutation rates, repeat
content, weights (small ones make ties, heavy ones overflow the packed cell form's fields) and overhang strategies.
usage: sw_fuzz.py SEEDS [FIRST_SEED]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import smithwaterman as sw  # noqa: E402
from oracle import oracle as O  # noqa: E402

WEIGHTS = [sw.ORIGINAL_DEFAULT, sw.STANDARD_NGS, (200, -150, -260, -11), (1, -1, -1, -1), (2, -2, -2, 0), (5, -4, -3, -3), (1, -3, -5, -2),
           (70000, -50000, -110000, -6000)]


def sequence(rng, n):
    kind = rng.integers(4)
    if kind == 0:  # a short repeat unit
        unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 6))))
        return (unit * (n // len(unit) + 1))[:n]
    if kind == 1:  # two-letter alphabet
        return "".join(rng.choice(list("AC"), size=n))
    return "".join(rng.choice(list("ACGTN") if kind == 2 else list("ACGT"), size=n))


def mutate(rng, s):
    s = list(s)
    for _ in range(int(rng.integers(0, 6))):
        s[rng.integers(len(s))] = "ACGT"[rng.integers(4)]
    for _ in range(int(rng.integers(0, 5))):
        if len(s) < 4:
            break
        p = int(rng.integers(1, len(s) - 1))
        k = int(rng.integers(1, 80 if rng.random() < 0.2 else 12))
        if rng.random() < 0.5:
            del s[p:p + k]
        else:
            s[p:p] = list(sequence(rng, k))
    return "".join(s) or "A"


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    pairs = 0
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(90000 + seed)
        refs, alts = [], []
        for _ in range(int(rng.integers(1, 60))):
            n = int(rng.choice([3, 20, 64, 130, 300, 450, 520, 1000, 1030])) + int(rng.integers(0, 40))
            ref = sequence(rng, n)
            lo, hi = int(rng.integers(0, n // 3 + 1)), n - int(rng.integers(0, n // 3 + 1))
            alt = mutate(rng, ref[lo:max(hi, lo + 1)] if rng.random() < 0.6 else ref)
            if rng.random() < 0.1:
                alt = sequence(rng, int(rng.integers(1, 700)))  # unrelated
            refs.append(ref.encode())
            alts.append(alt.encode())
        weights = WEIGHTS[int(rng.integers(len(WEIGHTS)))]
        strategy = int(rng.integers(4))
        got = sw.align_batch(refs, alts, weights, strategy, cigar_capacity=1024)
        for r, a, g in zip(refs, alts, got):
            rc, c, o = O.sw_align(r, a, weights, strategy)
            assert rc == 0 and g == (c, o), f"seed {seed}: {weights} strategy {strategy}\nref {r}\nalt {a}\ngot {g}\nwant {(c, o)}"
        pairs += len(refs)
        if seed % 20 == 19:
            print(f"seed {seed}: {pairs} alignments identical to the oracle, {time.time() - t0:.0f} s", flush=True)
    print(f"{seeds} seeds from {first}, {pairs} alignments: all identical to the oracle ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
